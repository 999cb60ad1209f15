cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=${1:-rX}; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu.txt 2>&1; tail -3 $O/${T}_pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh $T > /dev/null 2>&1
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('driver command: value %.4e ms_per_step %.4f kernel_avg_us %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us']))"; done
head -c 500 $O/${T}_bench.json; echo; head -8 $O/${T}_kernel_trace_stats.txt
