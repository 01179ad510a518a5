#!/bin/bash
# usage (on the GPU box): tools/timeline_cmd.sh <tag> <program and args (absolute paths)...>
# one rocprofv3 --kernel-trace run; the timeline of the last-but-one build goes to
# gpurun_out/timeline_<tag>.txt
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_tl_$tag
timeout -k 10 ${PASS_TIMEOUT:-240} rocprofv3 --kernel-trace -d /tmp/rp_tl_$tag -o r -- "$@" > "$out/timeline_$tag.log" 2>&1 || { echo "trace failed"; tail -3 "$out/timeline_$tag.log"; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/build_timeline.py $(find /tmp/rp_tl_$tag -name '*.db' | head -1) ${WHICH:--2} > "$out/timeline_$tag.txt" 2>>"$out/timeline_$tag.log"
rm -rf /tmp/rp_tl_$tag
