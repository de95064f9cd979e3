#!/bin/bash
# builds diagnostic variants of the library (wrong results by construction, timing only) into tools/ablate/
#   core:    the fused core without barriers, global loads and LDS traffic
#   strided: the column passes without their butterflies and twiddle multiplies (their memory floor)
#   math64:  the 64-bit tile and chunk passes without their arithmetic (their memory floor)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate
build() {  # name, defines...
  name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -x hip "$@" ntt_amd/csrc/ntt_api.cpp ntt_amd/csrc/ntt_textio.cpp -o tools/ablate/libnttamd_$name.so
}
build core -DNTT_ABLATE_BARRIER -DNTT_ABLATE_GLOAD -DNTT_ABLATE_LDS
build gload -DNTT_ABLATE_GLOAD
build strided -DNTT_ABLATE_STRIDED_MATH
build math64 -DNTT_ABLATE64_MATH -ffp-contract=off
ls -la tools/ablate
