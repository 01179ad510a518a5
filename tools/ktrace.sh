#!/bin/bash
# usage (on the GPU box): tools/ktrace.sh <tag> <program and args (absolute paths)...>
# ONE rocprofv3 --kernel-trace --stats run; the per-kernel summary goes to
# gpurun_out/ktrace_<tag>.csv (no PMC passes: see prof_cmd.sh for those).
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_kt_$tag
timeout -k 10 ${PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --stats -d /tmp/rp_kt_$tag -o r -- "$@" > "$out/ktrace_$tag.log" 2>&1 || { echo "trace failed"; tail -3 "$out/ktrace_$tag.log"; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/rocpd_kernel_stats.py $(find /tmp/rp_kt_$tag -name '*.db' | head -1) "$out/ktrace_$tag.csv" "$*" > /dev/null 2>>"$out/ktrace_$tag.log"
rm -rf /tmp/rp_kt_$tag
