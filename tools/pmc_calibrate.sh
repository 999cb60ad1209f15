#!/bin/bash
# Developer tool (GPU box): FETCH_SIZE / WRITE_SIZE of kernels with known byte counts -> gpurun_out/<tag>_pmc_calibration.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out/${T}_pmc_calibration.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calibrate tools/microbench/pmc_calibrate.hip 2>/dev/null
/tmp/pmc_calibrate > $O
for C in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/prof_cal; rm -rf $d
  rocprofv3 --pmc $C -d $d -o r -- /tmp/pmc_calibrate > /dev/null 2>&1
  python3 tools/rocprof_summary.py pmc $(find $d -name "*.db" | head -1) "cal_" >> $O
done
rm -rf gpurun_out/prof_cal
echo "# counters in KiB per dispatch; 64 rows x 65536 lanes x 4 B = 16384 KiB, the tile 65536 x 113 x 4 B = 28928 KiB" >> $O
cat $O
