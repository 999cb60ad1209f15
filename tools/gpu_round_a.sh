cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r4_b_pytest_gpu.txt 2>&1; tail -3 $O/r4_b_pytest_gpu.txt
bash tools/driver_repro.sh r4_b > /dev/null 2>&1; cat $O/r4_b_driver_repro.txt
python3 bench.py > $O/r4_b_bench.json 2>$O/r4_b_bench.err; head -c 600 $O/r4_b_bench.json
