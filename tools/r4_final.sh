#!/bin/bash
# round 4, final collection on one box: full -m gpu suite, fuzz sweeps, the bench line (timed), profiles of the
# headline run (C2), of the bf16 projection pass and of C3 / C4 / C5
mkdir -p gpurun_out/r4z
python -m pytest tests -m gpu -x -q > gpurun_out/r4z/pytest.log 2>&1; tail -3 gpurun_out/r4z/pytest.log
python tools/fuzz_parity.py 120 4411 > gpurun_out/r4z/fuzz_a.log 2>&1; tail -1 gpurun_out/r4z/fuzz_a.log
python tools/fuzz_parity.py 90 4412 heavy > gpurun_out/r4z/fuzz_b.log 2>&1; tail -1 gpurun_out/r4z/fuzz_b.log
t0=$(date +%s)
python bench.py > gpurun_out/r4z/bench.json 2> gpurun_out/r4z/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s" | tee gpurun_out/r4z/bench_wall.txt
export PMC_SETS="$(cat tools/pmc_sets_bench.txt)"
bash tools/prof_cmd.sh r04_main python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --other-configs none --shard-sweep none > gpurun_out/prof_r04_main.log 2>&1
export PMC_SETS="$(cat tools/pmc_sets_bf16.txt)"
bash tools/prof_cmd.sh r04_bf16 python3 $GRAFT_REPO_ROOT/tools/bf16_ab.py 10000000 768 128 0,8,3 > gpurun_out/prof_r04_bf16.log 2>&1
export PMC_SETS="FETCH_SIZE
WRITE_SIZE"
for c in c3 c4 c5; do bash tools/prof_cmd.sh r04_$c python3 $GRAFT_REPO_ROOT/bench.py --_other-child --other-configs $c --no-cpu-baseline --steps 3 > gpurun_out/prof_r04_$c.log 2>&1; done
ls gpurun_out/prof_r04_main gpurun_out/prof_r04_bf16 gpurun_out/prof_r04_c3 gpurun_out/prof_r04_c4 gpurun_out/prof_r04_c5
