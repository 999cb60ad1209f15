cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r2e_gemm_pmc.txt; : > $O
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/prof_gemm; rm -rf $d
  rocprofv3 --pmc $C -d $d -o r -- python3 tools/gemm_one.py fwd 8192 400 200 > /dev/null 2>&1
  python3 tools/rocprof_summary.py pmc $(find $d -name "*.db" | head -1) "k_gemm" >> $O
done
rm -rf gpurun_out/prof_gemm
cat $O
