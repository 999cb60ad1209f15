#!/bin/bash
# Developer tool (GPU box): SQ counter passes over bench.py's k_step (separate --pmc runs, no tracing combined).
#   tools/pmc_sq.sh <out-prefix>        writes gpurun_out/<prefix>_sq.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
P=${1:-sq}
OUT=gpurun_out/${P}_sq.txt
: > $OUT
run() {
  d=gpurun_out/pmc_$P/$1; shift
  rocprofv3 --pmc "$@" -d $d -o r -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  echo "# rocprofv3 --pmc $* -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline" >> $OUT
  python3 tools/rocprof_summary.py pmc $(find $d -name "*.db" | head -1) "k_env<9, false, true, 63" >> $OUT
  rm -rf $d
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU
run c SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VALU_TRANS_F32
run d SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQC_TC_STALL
cat $OUT
