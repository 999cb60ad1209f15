#!/bin/bash
# The 1 -> 8 GPU scaling curve of the BASELINE metric on ONE node with 8 MI355X, in one command (no such node has been available
# to a builder round: gpurun boxes have one GPU).  For N = 1, 2, 4, 8 it runs EXACTLY the driver's command line
#     python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
# (one rank per GPU, backend nccl = RCCL; bench.py prints ONE JSON line per run with both readings of the metric: `value` = weak scaling, 65536 envs per GPU,
# and `value_strong_65536_total` = the headline's 65536 envs partitioned over the ranks), then configs[3] (difficulty 4 + full domain randomisation,
# 16384 envs per GPU, the episode-statistics all-reduce every 4 steps) and configs[4] (PPO, 8192 envs per GPU, one gradient all-reduce per minibatch) at N = 8.
#     tools/scale_8gpu.sh [out_dir] [steps] [warmup]
set -u
cd "$(dirname "$0")/.." || exit 1
OUT=${1:-gpurun_out/scale8}; STEPS=${2:-200}; WARM=${3:-20}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NG=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NG" | tee "$OUT/scale.txt"
port=29610
for N in 1 2 4 8; do
    [ "$N" -le "$NG" ] || { echo "N=$N: skipped (only $NG GPUs)" | tee -a "$OUT/scale.txt"; continue; }
    port=$((port + 1))
    if [ "$N" -eq 1 ]; then
        python3 bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
    else
        python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$port" \
            bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
    fi
    python3 - "$OUT/bench_n$N.json" "$N" <<'PY' | tee -a "$OUT/scale.txt"
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line:
    print(f"N={sys.argv[2]}: no JSON line (see the .err file)"); sys.exit(0)
d = json.loads(line[0])
s = d.get("strong_scaling") or {}
print(f"N={d['n_gpus']}: weak {d['value']:.4e} env-steps/s ({d['ms_per_step'] * 1e3:.1f} us/step, 65536 envs/GPU) | strong 65536 total {s.get('value', float('nan')):.4e} "
      f"({s.get('ms_per_step', float('nan')) * 1e3:.1f} us/step, {s.get('envs_per_gpu')} envs/GPU, kernel {s.get('kernel_variant', '-')}) | fast contact set {d.get('value_fast_contact_set') or float('nan'):.4e}")
PY
done
if [ "$NG" -ge 8 ]; then
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29631 \
        bench.py --gpus 8 --steps "$STEPS" --warmup "$WARM" --envs 16384 --dr --stats-every 4 > "$OUT/bench_config3_n8.json" 2> "$OUT/bench_config3_n8.err"
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29632 \
        scripts/train_ppo.py gym=trifinger_difficulty_4 args.num_envs=8192 epochs=40 > "$OUT/ppo_config4_n8.txt" 2>&1
    tail -3 "$OUT/ppo_config4_n8.txt" | tee -a "$OUT/scale.txt"
fi
echo "scaling efficiency is the driver's to compute from the per-N values above: $OUT/scale.txt"
