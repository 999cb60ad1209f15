#!/bin/bash
# Developer tool (GPU box): frames/s of the in-repo PPO at the BASELINE configs[4] per-GPU shape, then its kernel trace.
#   tools/ppo_profile.sh <tag>  ->  gpurun_out/<tag>_ppo_rate.txt, gpurun_out/<tag>_ppo_kernel_trace.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}
O=gpurun_out
CMD="python3 scripts/train_ppo.py gym=trifinger_difficulty_4 args.num_envs=8192"
$CMD epochs=40 2>&1 | grep -v amdgpu.ids > $O/${T}_ppo_rate.txt
tail -4 $O/${T}_ppo_rate.txt
rocprofv3 --kernel-trace --stats -d $O/prof_ppo -o r -- $CMD epochs=8 > /dev/null 2>&1
{ echo "# command: rocprofv3 --kernel-trace --stats -- $CMD epochs=8"
  python3 tools/rocprof_summary.py trace $(find $O/prof_ppo -name "*.db" | head -1); } > $O/${T}_ppo_kernel_trace.txt
rm -rf $O/prof_ppo
head -32 $O/${T}_ppo_kernel_trace.txt | cut -c1-160
