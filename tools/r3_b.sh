cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 600 python tools/penetration_stats.py 65536 3000 2>&1 | grep -v amdgpu.ids > $O/r3_b_penetration.txt; cat $O/r3_b_penetration.txt
timeout 600 python tools/penetration_stats.py 65536 3000 dr 2>&1 | grep -v amdgpu.ids >> $O/r3_b_penetration.txt; tail -4 $O/r3_b_penetration.txt
