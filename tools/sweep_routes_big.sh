for n in 1024 4096 16384 65536; do for B in 1 16 256; do timeout -k 10 100 python tools/bench_routes_big.py $B $n 2>&1 | grep "n=" || exit 1; done; done
