#!/bin/bash
# Developer tool: compile the HIP source with --save-temps and print an opcode histogram of one kernel.
#   tools/isa_stats.sh [mangled-kernel-prefix]     (default: the headline k_step<9,false,true>)
set -e
K=${1:-_Z6k_stepILi9ELb0ELb1EE}
OUT=${ISA_OUT:-/tmp/dis}
mkdir -p "$OUT"
SRC=/root/repo/leibnizgym_amd/csrc/trifinger_hip.hip
(cd "$OUT" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
    -fno-slp-vectorize -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 --save-temps=obj -c -o "$OUT/tf.o" "$SRC" 2>&1 | grep -E "error|warning" -A3 || true)
S="$OUT/trifinger_hip-hip-amdgcn-amd-amdhsa-gfx950.s"
awk -v k="^$K[^ ]*:" '$0 ~ k {f=1} f{print} /^\.Lfunc_end/{if(f)exit}' "$S" > "$OUT/kernel.s"
echo "lines: $(wc -l < "$OUT/kernel.s")   ->  $OUT/kernel.s"
grep -E "^\s+\.(sgpr|vgpr|agpr)_count|scratch|\.private_segment_fixed_size|codeLenInByte|; (NumVgprs|NumAgprs|NumSgprs|ScratchSize|Occupancy)" "$OUT/kernel.s" | head -12
awk '{print $1}' "$OUT/kernel.s" | grep -v "^[.;]" | grep -v ":$" | sort | uniq -c | sort -rn | head -${ISA_TOP:-45}
