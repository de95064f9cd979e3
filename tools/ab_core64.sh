#!/bin/bash
# interleaved A/B of diagnostic builds of the fused 64-bit core (tools/ablate/libnttamd_core?.so) on one box
for round in 1 2; do
  for v in A B D E; do
    echo "variant $v round $round"
    timeout -k 10 100 python tools/profile_config4.py --lib tools/ablate/libnttamd_core$v.so 64 auto 2>&1 | grep -E "products/s|k64_core" || exit 1
  done
done
