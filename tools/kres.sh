#!/bin/bash
# register / spill / occupancy table of the kernels of one source file whose (demangled) name matches $2
# usage: tools/kres.sh csrc/project.hip proj_bf16x3
cd "$(dirname "$0")/../rp-tree_amd" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
  awk '/Function Name:/ {name=$5} /VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /VGPRs Spill:/ {s=$5} /Occupancy/ {o=$5} /LDS Size/ {print name, "vgpr", v, "agpr", a, "spill", s, "occ", o}' |
  while read -r n rest; do echo "$(echo "$n" | c++filt | sed 's/^void rpt::(anonymous namespace):://; s/(.*//') $rest"; done | grep -- "$2"
