# build / kNN time of one GPU's tree shard at the C2 shape, as bench.py --gpus N would see it
for t in 16 8 4; do
  echo "trees=$t"
  timeout -k 10 200 python bench.py --trees $t --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python tools/bench_summary.py
done
