#!/bin/bash
# build libpdt variants with different tile parameters into gpurun_out/variants (experiments only)
set -e
mkdir -p variants
i=0
for v in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function $v -Iinclude -shared -o variants/libpdt_$i.so project-desert-tortoise_amd/csrc/pdt_api.hip &
done
wait
