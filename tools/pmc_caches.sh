cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r2_l_pmc_caches.txt; : > $O
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_VALU"; do
  d=gpurun_out/prof_c; rm -rf $d
  rocprofv3 --pmc $C -d $d -o r -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/rocprof_summary.py pmc $(find $d -name "*.db" | head -1) "k_env<9, false, true, 127" >> $O 2>&1
done
rm -rf gpurun_out/prof_c; cat $O
