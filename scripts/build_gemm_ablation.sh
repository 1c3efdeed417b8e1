#!/bin/bash
# Ablation builds of the 256 x 256 x 64 GEMM K loop (gemm.hip, -DTRIBE_ABL_*): one GEMM-only library per variant under ab_tmp/
# (git-ignored, travels with the gpurun snapshot).  scripts/gemm_ablation.py times them interleaved in one process.
#   base      the kernel as shipped            nolds    no fragment ds_reads (LDS-DMA + MFMA + barriers)
#   nostage   no LDS-DMA (reads + MFMA)        bare     neither (MFMA + barriers)          nomfma   LDS-DMA + reads, no MFMA
set -e
cd "$(dirname "$0")/../algonauts-2025_amd/csrc"
mkdir -p ../../ab_tmp
FL="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function"
build() { /opt/rocm/bin/hipcc $FL $2 -c gemm.hip -o ../../ab_tmp/gemm_abl_$1.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 ../../ab_tmp/gemm_abl_$1.o abi.o -o ../../ab_tmp/libgemm_abl_$1.so; }
build base "" &
build nolds "-DTRIBE_ABL_NO_LDSREAD" &
build nostage "-DTRIBE_ABL_NO_STAGE" &
build bare "-DTRIBE_ABL_NO_LDSREAD -DTRIBE_ABL_NO_STAGE" &
wait
build nomfma "-DTRIBE_ABL_NO_MFMA" &
build dmaonly "-DTRIBE_ABL_NO_MFMA -DTRIBE_ABL_NO_LDSREAD" &
wait
ls -la ../../ab_tmp/libgemm_abl_*.so
