#!/bin/bash
# usage (GPU box): tools/c5_ab.sh <config> <opts1> <opts2> ...   — the other_configs child of bench.py once
# per option set (RPT_BENCH_OPTIONS syntax, "-" = defaults); prints build / split / kNN ms per set
cfg=$1; shift
for o in "$@"; do
  [ "$o" = "-" ] && o=""
  RPT_BENCH_OPTIONS="$o" python bench.py --_other-child --other-configs $cfg --steps 3 --no-cpu-baseline > gpurun_out/ab_tmp.log 2>gpurun_out/ab_tmp.err || { echo "failed: $o"; tail -3 gpurun_out/ab_tmp.err; continue; }
  python - "$o" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.log").read().strip().splitlines()[-1])
for k, r in d.items():
    if isinstance(r, dict) and "build_ms" in r:
        print("%-24s %s build %.2f ms (projection %.2f, split %.2f)  knn %.2f ms" % (
            sys.argv[1] or "(defaults)", k, r["build_ms"], r["build_breakdown_ms"]["projection_total"],
            r["build_breakdown_ms"]["split_total"], r["knn_ms_per_batch"]), flush=True)
PY
done
