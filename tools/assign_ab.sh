#!/bin/bash
# stream_assign blocks per CU (option tune0): C2 whole, a 4-tree shard, the C4 shard
mkdir -p gpurun_out/r4w; rm -f gpurun_out/r4w/asg_*
for v in 1 2 1 2; do
  export RPT_TUNE0=$v
  timeout -k 10 200 python tools/shard4.py 32 30 >> gpurun_out/r4w/asg_c2_$v.log 2>&1 || exit 1
  timeout -k 10 200 python tools/shard4.py 4 50 >> gpurun_out/r4w/asg_c2_$v.log 2>&1 || exit 1
  timeout -k 10 300 python bench.py --_other-child --other-configs c4 --no-cpu-baseline --steps 5 >> gpurun_out/r4w/asg_c4_$v.json 2>> gpurun_out/r4w/asg_c4_$v.err || exit 1
done
