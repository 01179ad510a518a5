#!/bin/bash
# usage (on the GPU box): tools/prof_cmd.sh <tag> <program and args...>
# kernel trace + PMC passes of one command; rocpd databases stay in /tmp, only the per-kernel
# summaries (CSV) go to gpurun_out/prof_<tag>/ so that the merge back stays small.
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprofv3 args...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  echo "[prof] pass $name: $*"
  timeout -k 10 ${PASS_TIMEOUT:-240} rocprofv3 "$@" -d /tmp/rp_$name -o r -- "${CMD[@]}" > "$out/$name.log" 2>&1 || { echo "$name failed"; tail -3 "$out/$name.log"; return 1; }
}
CMD=("$@")
run trace --kernel-trace --stats || exit 1
python3 $GRAFT_REPO_ROOT/tools/rocpd_kernel_stats.py $(find /tmp/rp_trace -name '*.db' | head -1) "$out/kernel_stats.csv" "$*" > /dev/null 2>>"$out/trace.log"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  run pmc$i --kernel-trace --pmc $line || continue
  python3 $GRAFT_REPO_ROOT/tools/pmc_mean.py $(find /tmp/rp_pmc$i -name '*.db' | head -1) "$out/pmc$i.csv" $line > /dev/null 2>>"$out/pmc$i.log"
  rm -rf /tmp/rp_pmc$i
done <<< "${PMC_SETS:-FETCH_SIZE WRITE_SIZE}"
rm -rf /tmp/rp_trace
ls -la "$out"
