#!/bin/bash
# round 4, final collection on one box: full -m gpu suite, fuzz sweeps, the bench line, C3 / C4 / C5 profiles
mkdir -p gpurun_out/r4z
python -m pytest tests -m gpu -x -q > gpurun_out/r4z/pytest.log 2>&1; tail -3 gpurun_out/r4z/pytest.log
python tools/fuzz_parity.py 120 4401 > gpurun_out/r4z/fuzz_a.log 2>&1; tail -1 gpurun_out/r4z/fuzz_a.log
python tools/fuzz_parity.py 90 4402 heavy > gpurun_out/r4z/fuzz_b.log 2>&1; tail -1 gpurun_out/r4z/fuzz_b.log
python bench.py > gpurun_out/r4z/bench.json 2> gpurun_out/r4z/bench.err; echo "bench rc=$?"
export PMC_SETS="FETCH_SIZE
WRITE_SIZE"
for c in c3 c4 c5; do bash tools/prof_cmd.sh r04_$c python3 $GRAFT_REPO_ROOT/bench.py --_other-child --other-configs $c --no-cpu-baseline --steps 3 > gpurun_out/prof_r04_$c.log 2>&1; done
ls gpurun_out/prof_r04_c3 gpurun_out/prof_r04_c4 gpurun_out/prof_r04_c5
