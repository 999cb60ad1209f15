cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
T=r6_zz; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu.txt 2>&1; tail -3 $O/${T}_pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/${T}_pytest_gpu.txt
bash tools/profile_ext.sh $T > /dev/null 2>&1
python3 bench.py --difficulty 1 --envs 8192 --no-cpu-baseline > $O/${T}_bench_config1.json 2>/dev/null
{ timeout 600 python tools/soak.py 8192 200000; timeout 600 python tools/soak.py 16384 100000 box; } 2>&1 | grep -v amdgpu.ids > $O/${T}_soak_helpers.txt; tail -3 $O/${T}_soak_helpers.txt
for f in $O/${T}_ext_*_bench.json $O/${T}_bench_config1.json; do python3 - $f <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().splitlines()[0]); r = d["roofline"]
print(sys.argv[1], "value %.4e ms %.4f kern %.2f %s fast %s" % (d["value"], d["ms_per_step"], r["kernel_avg_us"], r["kernel_variant"], d.get("value_fast_contact_set")))
PY
done
grep "k_env" $O/${T}_ext_*_kernel_trace.txt | cut -c1-200
