#!/bin/bash
# builds diagnostic variants of the library (wrong results by construction, timing only)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate
# BARRIER+GLOAD+LDS: the fused core without barriers, global loads and LDS traffic; STRIDED_MATH: the column
# passes without their butterflies and twiddle multiplies (their memory floor)
for v in "BARRIER -DNTT_ABLATE_GLOAD -DNTT_ABLATE_LDS" "STRIDED_MATH"; do
  name=$(echo $v | tr -d ' -_' | sed 's/DNTTABLATE//g')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -x hip -DNTT_ABLATE_$v ntt_amd/csrc/ntt_api.cpp ntt_amd/csrc/ntt_textio.cpp -o tools/ablate/libnttamd_$name.so
done
ls -la tools/ablate
