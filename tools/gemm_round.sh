cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
for L in leibnizgym_amd/csrc/libtrifinger_ppo.so leibnizgym_amd/csrc/variants/libppo_gd2.so leibnizgym_amd/csrc/variants/libppo_gd4.so leibnizgym_amd/csrc/variants/libppo_minb2.so; do
  for CH in 256 512 1024; do echo "## $L CHUNK=$CH"; PPO_LIB=$L CHUNK=$CH python3 tools/gemm_kernels_bench.py 2>&1 | grep -v "amdgpu.ids\|rel err"; done
done
