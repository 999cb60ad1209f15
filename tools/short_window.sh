cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
for S in 20 20 50 100 200 500 2000; do python3 bench.py --gpus 1 --steps $S --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('steps %5d: value %.4e ms_per_step %.4f kernel_avg_us %.2f (%d launches timed)' % (d['steps'], d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us'], d['roofline']['kernel_launches_timed']))"; done
for W in 5 50 500; do python3 bench.py --gpus 1 --steps 20 --warmup $W --settle 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('settle 0 warmup $W steps 20: value %.4e ms_per_step %.4f kernel_avg_us %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us']))"; done
