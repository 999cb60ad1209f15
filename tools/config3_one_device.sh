#!/bin/bash
# Developer tool (GPU box with ONE GPU): BASELINE configs[3] at its full global size through the multi-process path - 131072 envs = 8 ranks x 16384, every
# domain-randomisation feature, episode statistics all-reduced every 4 steps - with all eight ranks on cuda:0 (TF_BENCH_SINGLE_DEVICE_TEST=1: collectives
# through gloo).  The eight ranks share one GPU: the rate is not a scaling number, the line shows that the path runs at that size.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
TF_BENCH_SINGLE_DEVICE_TEST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29571 \
  bench.py --gpus 8 --dr --envs 16384 --steps 200 --warmup 5 --stats-every 4 2>/dev/null | grep "^{" > $O/${1:-rX}_config3_8_ranks_one_device.json
python3 -c "
import json,sys
d=json.load(open('$O/${1:-rX}_config3_8_ranks_one_device.json'))
print('n_gpus', d['n_gpus'], 'global envs', d['config']['global_envs'], 'value %.4e'%d['value'], 'ms_per_step %.4f'%d['ms_per_step'], 'stats', [round(x,3) for x in d['episode_stats_all_reduced'][:11]])"
