cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r3_a_pytest_gpu.txt 2>&1; tail -3 $O/r3_a_pytest_gpu.txt
timeout 600 python tools/two_stream.py 2>&1 | grep -v amdgpu.ids > $O/r3_a_two_stream.txt; cat $O/r3_a_two_stream.txt
timeout 1500 bash tools/profile_ext.sh r3_a > $O/r3_a_ext_log.txt 2>&1; tail -8 $O/r3_a_ext_log.txt
for S in 7 11 23; do timeout 400 python tools/ppo_learning_check.py 800 8192 $S fused 4 2>&1 | grep -v amdgpu.ids > $O/r3_a_ppo_learning_d4_seed$S.txt; tail -12 $O/r3_a_ppo_learning_d4_seed$S.txt; done
