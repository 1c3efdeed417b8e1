#!/bin/bash
# Diagnostic copy of the library with s_memtime stamps in the attention kernels (-DTRIBE_ATTN_STAMPS) -> ab_tmp/ (git-ignored,
# travels with the gpurun snapshot).  Run in the container after `make` in csrc/.
set -e
cd "$(dirname "$0")/../algonauts-2025_amd/csrc"
mkdir -p ../../ab_tmp
for f in attention attention_d64; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -DTRIBE_ATTN_STAMPS -c $f.hip -o ../../ab_tmp/${f}_stamps.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 ../../ab_tmp/attention_stamps.o ../../ab_tmp/attention_d64_stamps.o \
  gemm.o elementwise.o loss.o encoder.o extractors.o backward.o features.o gemm_fp8.o abi.o -o ../../ab_tmp/libtribe_hip_attn_stamps.so
