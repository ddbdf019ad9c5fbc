#!/bin/bash
# builds ablation variants of the library (PSDR_ABL bitmask) into build/variants/
set -e
cd "$(dirname "$0")/.."
for abl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -shared -fPIC -DPSDR_ABL=$abl \
     -o build/variants/libpsdr_abl$abl.so phantomsdr_amd/csrc/psdr_api.hip &
done
wait
ls -la build/variants
