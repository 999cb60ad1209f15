cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1200 python -m pytest tests/test_ppo_kernels.py tests/test_ppo.py -x -q -m gpu 2>&1 | tail -3
python3 scripts/train_ppo.py gym=trifinger_difficulty_4 args.num_envs=8192 epochs=40 2>&1 | grep -v amdgpu.ids | tail -3
