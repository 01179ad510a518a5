"""Run a script under a faulthandler watchdog: dumps every thread's Python stack to stderr
and exits if the script has not finished after WATCHDOG_S seconds (default 120).
usage: python tools/run_with_watchdog.py script.py [args...]"""
import faulthandler
import os
import runpy
import sys

faulthandler.dump_traceback_later(int(os.environ.get("WATCHDOG_S", "120")), exit=True)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
faulthandler.cancel_dump_traceback_later()
print("watchdog: script returned", file=sys.stderr, flush=True)
faulthandler.dump_traceback_later(30, exit=True)   # interpreter shutdown
