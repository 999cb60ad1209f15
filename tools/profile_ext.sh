#!/bin/bash
# Developer tool (GPU box): time and profile what BASELINE configs[3] really launches - the EXT instantiation of the fused step
# (every domain-randomisation feature incl. base / stage offsets and per-body friction) - and the box-object variant.
#   tools/profile_ext.sh <tag>  ->  gpurun_out/<tag>_ext_{dr,box}_<N>_{bench.json,kernel_trace.txt,pmc.txt}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out
mkdir -p $O
for W in dr box; do
  for N in 16384 65536; do
    # <= 32768 envs: the 256-register instantiation; <= 16384 envs and not the box object: with helper wavefronts
    K="k_env<9, false, true, 127, $([ $W = box ] && echo 2 || echo 1), $([ $N -le 32768 ] && echo true || echo false), $([ $N -le 16384 ] && [ $W != box ] && echo true || echo false)>"
    B="python3 bench.py --$W --envs $N --no-cpu-baseline --no-fast-contact-leg"
    P=$O/${T}_ext_${W}_${N}
    rocprofv3 --kernel-trace --stats -d $O/prof_$T/trace -o r -- $B --steps 300 --warmup 5 > /dev/null 2>&1
    { echo "# command: rocprofv3 --kernel-trace --stats -- $B --steps 300 --warmup 5   (MI355X)"
      python3 tools/rocprof_summary.py trace $(find $O/prof_$T/trace -name "*.db" | head -1); } > ${P}_kernel_trace.txt
    rm -rf $O/prof_$T/trace
    { echo "# workload: N=$N asym=True kernel=$K ($W)"; } > ${P}_pmc.txt
    for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
      d=$O/prof_$T/pmc; rm -rf $d
      rocprofv3 --pmc $C -d $d -o r -- $B --steps 60 --warmup 5 > /dev/null 2>&1
      { echo "# command: rocprofv3 --pmc $C -- $B --steps 60 --warmup 5"
        python3 tools/rocprof_summary.py pmc $(find $d -name "*.db" | head -1) "k_env"; echo; } >> ${P}_pmc.txt
    done
    echo "# units: FETCH_SIZE / WRITE_SIZE in KiB per dispatch (raw rocprofv3 expressions); SQ cycle counters in quad-cycles summed over waves" >> ${P}_pmc.txt
    rm -rf $O/prof_$T
    $B > ${P}_bench.json 2>/dev/null
  done
done
head -c 400 $O/${T}_ext_dr_16384_bench.json; echo; grep "k_env" $O/${T}_ext_*_kernel_trace.txt
