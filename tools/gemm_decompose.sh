#!/bin/bash
# Developer tool (GPU box): what limits the trainer's tile kernel.  Builds csrc/ppo_kernels.hip with GEMM_DBG = 1, 2, 4, 5 (timing-only variants, see
# the source), times the six layer shapes with each, and runs the two microbenchmarks of tools/microbench/ (MFMA against another wavefront's vector /
# LDS work on the same SIMD).    tools/gemm_decompose.sh <tag>  ->  gpurun_out/<tag>_gemm_kernels.txt, gpurun_out/<tag>_mfma_overlap.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; V=leibnizgym_amd/csrc/variants; mkdir -p $O $V
for d in 1 2 4 5; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -DGEMM_DBG=$d -o $V/libppo_dbg$d.so leibnizgym_amd/csrc/ppo_kernels.hip 2>/dev/null; done
{ echo "# tools/gemm_decompose.sh: per call inside a HIP graph (20 calls per graph), M = 8192; 'mfma' = this repo's kernel, 'torch' = rocBLAS + elementwise"
  echo "## product build (leibnizgym_amd/csrc/libtrifinger_ppo.so)"; python3 tools/gemm_kernels_bench.py 2>&1 | grep "mfma"
  for d in 1 2 4 5; do
    case $d in 1) W="no MFMAs, no LDS reads: the staging side alone";; 2) W="no global loads inside the K loop";; 4) W="no LDS reads (MFMAs on constants)";; 5) W="staging wavefronts keep only the barriers: the multiplying side alone";; esac
    echo "## GEMM_DBG=$d ($W)"; PPO_LIB=$V/libppo_dbg$d.so SHAPES=400x200,113x400 python3 tools/gemm_kernels_bench.py 2>&1 | grep "mfma" | grep -v "^sum"
  done
  echo "## one K tile (K = 32): the fixed part of a launch"; SHAPES=32x200 python3 tools/gemm_kernels_bench.py 2>&1 | grep "mfma" | grep -v "^sum"
} > $O/${T}_gemm_kernels.txt
( cd tools/microbench && hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip 2>/dev/null )
{ echo "# tools/microbench/mfma_valu_overlap: 256 / 512 workgroups of 8 wavefronts; per SIMD one wavefront runs a v_mfma_f32_32x32x2_f32 chain (| variant), one runs 'work'"
  timeout 120 tools/microbench/mfma_valu_overlap; } > $O/${T}_mfma_overlap.txt 2>&1
cat $O/${T}_gemm_kernels.txt | cut -c1-100
