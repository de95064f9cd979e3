#!/bin/bash
# builds diagnostic variants of the library (wrong results by construction, timing only)
set -e
cd "$(dirname "$0")/.."
for v in "BARRIER -DNTT_ABLATE_GLOAD -DNTT_ABLATE_LDS"; do
  name=$(echo $v | tr -d ' -' | sed 's/DNTT_ABLATE_//g')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip -DNTT_ABLATE_$v ntt_amd/csrc/ntt_api.cpp -o tools/ablate/libnttamd_$name.so
done
ls -la tools/ablate
