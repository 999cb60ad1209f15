#!/bin/bash
# Developer tool (GPU box): the closing batch of round 6 in one gpurun call.   tools/closing_r6.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu.txt 2>&1; tail -3 $O/${T}_pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/${T}_pytest_gpu.txt
bash tools/profile_round.sh $T > $O/${T}_profile_log.txt 2>&1
bash tools/profile_ext.sh $T > /dev/null 2>&1
python3 tools/sweep.py 2>&1 | grep -v amdgpu.ids > $O/${T}_sweep.txt; cat $O/${T}_sweep.txt
python3 -c "from leibnizgym_amd import _capi; _capi.TfLib('leibnizgym_amd/csrc/libtrifinger_hip_timing.so')" 2>/dev/null || echo "STALE timing build: run make -C leibnizgym_amd/csrc libtrifinger_hip_timing.so before this script (the phase tables below will be an error trace)"
{ python3 tools/phase_timing.py 65536 600; python3 tools/phase_timing.py 8192 600; } 2>&1 | grep -v amdgpu.ids > $O/${T}_phase_timing_steady.txt
python3 tools/api_bench.py 2>&1 | grep -v amdgpu.ids > $O/${T}_api_layers.txt
python3 bench.py --difficulty 1 --envs 8192 --no-cpu-baseline > $O/${T}_bench_config1.json 2>/dev/null
# the multi-rank command lines on one device (plumbing, not rates): 2 and 8 ranks with the strong-scaling leg
TF_BENCH_SINGLE_DEVICE_TEST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 8 --steps 50 --warmup 5 --envs 8192 --settle 100 --strong-total 65536 2>/dev/null | grep "^{" > $O/${T}_bench_8_ranks_one_device.json
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('driver command: value %.4e ms_per_step %.4f kernel_avg_us %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us']))"; done | tee $O/${T}_driver_command.txt
timeout 900 python tools/soak.py 65536 100000 2>&1 | grep -v amdgpu.ids > $O/${T}_soak.txt; tail -2 $O/${T}_soak.txt
timeout 600 python tools/soak.py 65536 20000 box 2>&1 | grep -v amdgpu.ids > $O/${T}_soak_box.txt; tail -1 $O/${T}_soak_box.txt
{ timeout 600 python tools/penetration_stats.py 65536 3000; timeout 600 python tools/penetration_stats.py 65536 3000 dr; } 2>&1 | grep -v amdgpu.ids > $O/${T}_penetration.txt; cat $O/${T}_penetration.txt
timeout 500 python tools/ppo_learning_check.py 800 8192 7 fused 4 2>&1 | grep -v amdgpu.ids > $O/${T}_ppo_learning_d4_seed7.txt; grep "play step  700" $O/${T}_ppo_learning_d4_seed7.txt | head -1 | cut -c1-260
bash tools/ppo_profile.sh $T > /dev/null 2>&1; tail -3 $O/${T}_ppo_rate.txt
python3 tools/wall_census.py 2>&1 | grep -v amdgpu.ids > $O/${T}_wall_census.txt; tail -1 $O/${T}_wall_census.txt
head -c 500 $O/${T}_bench.json; echo; head -6 $O/${T}_kernel_trace_stats.txt
# round 6: the step of both contact sets over N (default model = reference's self-collision set; FF_MIDDLE=0 = the fast set of API <= 7), the trainer's products
{ ASYM=1 python3 tools/variant_sweep.py 8192 16384 32768 65536; ASYM=1 FF_MIDDLE=0 python3 tools/variant_sweep.py 8192 16384 32768 65536 | sed "s/^/fast contact set: /"; } 2>&1 | grep -v amdgpu.ids > $O/${T}_variant_sweep.txt; cat $O/${T}_variant_sweep.txt
{ python3 tools/experiments/walk_bench.py; python3 tools/experiments/dw_direct_bench.py; } 2>&1 | grep -v amdgpu.ids > $O/${T}_trainer_products.txt; head -4 $O/${T}_trainer_products.txt
