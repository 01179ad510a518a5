#!/bin/bash
# usage (on the GPU box): tools/mfma_clock.sh [waves per SIMD = 2] [iterations = 20000]
# The shader clock a chain-free stream of v_mfma_f64_16x16x4_f64 actually runs at: one rocprofv3
# pass with GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) next to the kernel's duration.
set -u
W=${1:-2}; IT=${2:-20000}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p "$out"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $GRAFT_REPO_ROOT/tools/mfma_f64_peak.hip -o /tmp/mfma_f64_peak || exit 1
/tmp/mfma_f64_peak $W $IT | tee "$out/mfma_clock.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_mc
timeout -k 10 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/rp_mc -o r -- /tmp/mfma_f64_peak $W $IT > "$out/mfma_clock.log" 2>&1 || { echo "pmc pass failed"; tail -3 "$out/mfma_clock.log"; exit 1; }
python3 - "$(find /tmp/rp_mc -name '*.db' | head -1)" <<'PY' | tee -a "$out/mfma_clock.txt"
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
dur = {}
for name, s, e in db.execute("select name, start, end from kernels order by start"):
    dur.setdefault(name, []).append(e - s)
cyc = {}
for name, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
    if cname == "GRBM_GUI_ACTIVE":
        cyc.setdefault(name, []).append(val)
for name in cyc:
    d, c = dur.get(name, [0])[-1], cyc[name][-1]
    if d:
        print("%s: %.3f ms, GRBM_GUI_ACTIVE %.4g (8 XCDs) -> %.0f MHz shader clock under this kernel" %
              (name[:40], d / 1e6, c, c / 8.0 / (d / 1e9) / 1e6))
PY
rm -rf /tmp/rp_mc
