#!/bin/bash
# Developer tool (GPU box): trace and counters of the step kernel at the per-GPU sizes of BASELINE configs[1] / [4] (8192 envs) and [3] (16384 envs) - the
# 256-register kernel with helper wavefronts.   tools/profile_small_n.sh <tag>  ->  gpurun_out/<tag>_small_n_{kernel_trace,pmc}.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; mkdir -p $O
: > $O/${T}_small_n_kernel_trace.txt; : > $O/${T}_small_n_pmc.txt
for N in 8192 16384; do
  B="python3 bench.py --envs $N --no-cpu-baseline --no-fast-contact-leg"
  rocprofv3 --kernel-trace --stats -d $O/prof_sn/trace -o r -- $B --steps 500 --warmup 5 > /dev/null 2>&1
  { echo "# command: rocprofv3 --kernel-trace --stats -- $B --steps 500 --warmup 5   (MI355X)"
    python3 tools/rocprof_summary.py trace $(find $O/prof_sn/trace -name "*.db" | head -1) | head -8; echo; } >> $O/${T}_small_n_kernel_trace.txt
  rm -rf $O/prof_sn/trace
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
    d=$O/prof_sn/pmc; rm -rf $d
    rocprofv3 --pmc $C -d $d -o r -- $B --steps 60 --warmup 5 > /dev/null 2>&1
    { echo "# N=$N command: rocprofv3 --pmc $C -- $B --steps 60 --warmup 5"
      python3 tools/rocprof_summary.py pmc $(find $d -name "*.db" | head -1) "k_env<9, false, true, 127"; echo; } >> $O/${T}_small_n_pmc.txt
  done
  rm -rf $O/prof_sn
done
cat $O/${T}_small_n_kernel_trace.txt | cut -c1-170
