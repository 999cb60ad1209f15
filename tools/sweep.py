"""Throughput sweep over env counts on one GPU (BASELINE.md section 4 'sweep'); prints a small table."""
import json, subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for sym in (False, True):
    for n in (8192, 16384, 32768, 65536, 131072, 262144):
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--envs", str(n), "--steps", "500", "--warmup", "10", "--no-cpu-baseline"]
        if sym: cmd.append("--symmetric")
        out = subprocess.run(cmd, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        rows.append((n, not sym, d["value"], d["ms_per_step"], d["roofline"]["kernel_avg_us"], d["roofline"]["frac"]))
        print(f"N={n:7d} asym={not sym!s:5s} {d['value']:.4e} env-steps/s  {d['ms_per_step']*1e3:8.1f} us/step  k_env {d['roofline']['kernel_avg_us']:7.1f} us ({d['roofline'].get('kernel_variant', '?'):6s})  hbm frac {d['roofline']['frac']:.4f}", flush=True)
