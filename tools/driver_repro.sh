#!/bin/bash
# Developer tool (GPU box): the driver's exact bench command in fresh processes, next to longer warm-ups / runs, and a per-launch
# kernel trace of the driver command (duration of every k_env launch in order: clock ramp vs workload).
#   tools/driver_repro.sh <tag>   ->  gpurun_out/<tag>_driver_repro.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
set +e
T=${1:-rX}; O=gpurun_out; mkdir -p $O
F=$O/${T}_driver_repro.txt
pick() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('value %.4e  ms_per_step %.4f  kernel_avg_us %.2f (%d launches timed)  resident_actions %.4e' % (d['value'], d['ms_per_step'], r['kernel_avg_us'], r['kernel_launches_timed'], d['value_resident_actions']))
"; }
{
echo "# rocm-smi clocks before"; rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk|fclk" | head -4
for i in 1 2 3 4 5 6; do echo "# run $i: python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline (fresh process)"; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | pick; done
echo "# the driver's command as is (with the cpu_baseline leg)"; python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | pick
for W in 100 500 2000; do echo "# --steps 20 --warmup $W"; python3 bench.py --gpus 1 --steps 20 --warmup $W --no-cpu-baseline 2>/dev/null | pick; done
for S in 200 2000; do echo "# --steps $S --warmup 10"; python3 bench.py --gpus 1 --steps $S --warmup 10 --no-cpu-baseline 2>/dev/null | pick; done
echo "# per-launch durations of k_env under rocprofv3 --kernel-trace, driver command (us, launch order: reset, 5 warm-up, 20 timed, ...)"
rm -rf $O/prof_dr; rocprofv3 --kernel-trace -d $O/prof_dr -o r -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import sqlite3, glob
db = glob.glob("$O/prof_dr/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels where name like '%k_env%' order by start").fetchall()
t0 = rows[0][1]
print(" ".join("%s@%.0f:%.1f" % ("R" if "<9, true" in n else "S", (s - t0) / 1e3, (e - s) / 1e3) for n, s, e in rows[:40]))
PY
rm -rf $O/prof_dr
echo "# rocm-smi clocks after"; rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk|fclk" | head -4
} > $F 2>&1
cat $F
