#!/bin/bash
# short large-modulus products: shipped build against a diagnostic one, interleaved
for n in 64 1024 2048; do for B in 256 4096; do
  timeout -k 10 100 python tools/bench_routes_big.py $B $n 2>&1 | grep "64-bit field" || exit 1
  timeout -k 10 100 python tools/bench_routes_big.py --lib tools/ablate/libnttamd_shortv2.so $B $n 2>&1 | grep "64-bit field" | sed 's/^/   v2: /' || exit 1
done; done
