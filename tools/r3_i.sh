cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=r3_i; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_hip_vs_oracle.py tests/test_box_object.py tests/test_domain_randomization.py -m gpu -x -q > $O/${T}_pytest_gpu.txt 2>&1; tail -2 $O/${T}_pytest_gpu.txt
bash tools/profile_round.sh $T > $O/${T}_profile_log.txt 2>&1
bash tools/profile_ext.sh $T > $O/${T}_ext_log.txt 2>&1
for f in dr_16384 dr_65536 box_16384 box_65536; do head -c 200 $O/${T}_ext_${f}_bench.json | cut -c100-200; echo; done
head -c 300 $O/${T}_bench.json; echo
