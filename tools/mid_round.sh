#!/bin/bash
# Developer tool (GPU box): GPU suite + smoke, the headline profile entry, the N sweep and the phase tables of both step-kernel instantiations.
#   tools/mid_round.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu.txt 2>&1; tail -3 $O/${T}_pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/${T}_pytest_gpu.txt
bash tools/profile_round.sh $T > $O/${T}_profile_log.txt 2>&1
python3 tools/sweep.py 2>&1 | grep -v amdgpu.ids > $O/${T}_sweep.txt
{ python3 tools/phase_timing.py 65536 600; python3 tools/phase_timing.py 8192 600; VARIANT=narrow python3 tools/phase_timing.py 8192 600; } 2>&1 | grep -v amdgpu.ids > $O/${T}_phase_timing_steady.txt
python3 tools/wall_census.py 2>&1 | grep -v amdgpu.ids > $O/${T}_wall_census.txt
cat $O/${T}_sweep.txt; head -c 600 $O/${T}_bench.json; echo; tail -2 $O/${T}_wall_census.txt
