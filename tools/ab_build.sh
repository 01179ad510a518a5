# A/B of two builds of the library on one box: bash tools/ab_build.sh ab/lib_a.so ab/lib_b.so [bench args]
A=$1; B=$2; shift 2
for rep in 1 2; do
  for lib in $A $B; do
    echo -n "$(basename $lib): "
    RPTREE_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python tools/bench_summary.py | cut -c1-150
  done
done
