#!/bin/bash
# Developer tool (GPU box): hardware counters of the direct weight-gradient kernel (L2 hit rate, MFMA busy), separate rocprofv3 --pmc passes
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
cat > /tmp/dw_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from leibnizgym_amd import ppo_kernels as pk
dev = "cuda:0"; rows = 8192
shapes = [(400, 41), (200, 400), (100, 200), (9, 100), (400, 113), (200, 400), (100, 200), (1, 100)]
as_ = [torch.randn(rows, n1, device=dev) for n1, _ in shapes]; bs = [torch.randn(rows, n2, device=dev) for _, n2 in shapes]
outs = [(torch.zeros(n1, n2, device=dev), torch.zeros(n1, device=dev)) for n1, n2 in shapes]
for _ in range(20):
    pk.gemm_tn_bias_direct(as_, bs, outs); pk.discard_partial_sums()
torch.cuda.synchronize()
PY
{ for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  d=$O/prof_dw; rm -rf $d
  rocprofv3 --pmc $C -d $d -o r -- python3 /tmp/dw_once.py > /dev/null 2>&1
  echo "# rocprofv3 --pmc $C"; python3 tools/rocprof_summary.py pmc $(find $d -name "*.db" | head -1) "k_dw_direct"; echo
done; rm -rf $O/prof_dw; } > $O/dw_pmc.txt 2>&1
cat $O/dw_pmc.txt | cut -c1-180
