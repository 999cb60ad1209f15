#!/bin/bash
# Developer tool (GPU box): everything a round's profiles/ entry needs, in one gpurun call.
#   tools/profile_round.sh <tag>   ->  gpurun_out/<tag>_{bench.json,bench_symmetric.json,kernel_trace_stats.txt,pmc.txt}
# Tracing and counters are separate rocprofv3 runs; the profiled program is python3 itself (no launcher hop).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}
O=gpurun_out
K='k_env<9, false, true, 127, 0, false, false>'
mkdir -p $O
CMD="python3 bench.py --steps 500 --warmup 5 --no-cpu-baseline --no-fast-contact-leg"
rocprofv3 --kernel-trace --stats -d $O/prof_$T/trace -o r -- $CMD > /dev/null 2>&1
{ echo "# command: rocprofv3 --kernel-trace --stats -- $CMD   (MI355X, 65536 envs, asymmetric obs)"
  python3 tools/rocprof_summary.py trace $(find $O/prof_$T/trace -name "*.db" | head -1); } > $O/${T}_kernel_trace_stats.txt
CMD="python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-fast-contact-leg"
{ echo "# workload: N=65536 asym=True kernel=$K"; } > $O/${T}_pmc.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  d=$O/prof_$T/pmc
  rm -rf $d
  rocprofv3 --pmc $C -d $d -o r -- $CMD > /dev/null 2>&1
  { echo "# command: rocprofv3 --pmc $C -- $CMD"
    python3 tools/rocprof_summary.py pmc $(find $d -name "*.db" | head -1) "k_env"; echo; } >> $O/${T}_pmc.txt
done
echo "# units: FETCH_SIZE / WRITE_SIZE in KiB per dispatch (raw rocprofv3 expressions); SQ cycle counters in quad-cycles summed over waves" >> $O/${T}_pmc.txt
rm -rf $O/prof_$T
# the headline bench lines AFTER the counters exist under profiles/ is the normal order; here the fresh file is used directly
mkdir -p profiles && cp $O/${T}_pmc.txt profiles/${T}_pmc.txt
python3 bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
python3 bench.py --symmetric --no-cpu-baseline > $O/${T}_bench_symmetric.json 2>> $O/${T}_bench.err
cat $O/${T}_bench.json $O/${T}_kernel_trace_stats.txt
