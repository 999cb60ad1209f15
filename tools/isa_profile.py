#!/usr/bin/env python3
"""Developer tool: static instruction profile of one kernel by source function (optionally focused on one function).
Compiles the HIP source with line tables, walks the kernel's ISA and attributes every instruction to the source line of
its innermost .loc, then sums per function (line ranges taken from the DEV / __global__ definitions in the source).
    python tools/isa_profile.py [mangled-kernel-prefix [focus-function]]"""
import bisect, collections, os, re, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "leibnizgym_amd", "csrc", "trifinger_hip.hip")
K = sys.argv[1] if len(sys.argv) > 1 else "_Z6k_stepILi9ELb0ELb1EE"
FOCUS = sys.argv[2] if len(sys.argv) > 2 else None      # optional: only instructions inlined (at any depth) from this function
OUT = "/tmp/dis_prof"
os.makedirs(OUT, exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                       "-fno-fast-math", "-fno-slp-vectorize", "-Wno-unused-function", "-gline-tables-only",
                       "--save-temps=obj", "-c", "-o", OUT + "/tf.o", SRC], cwd=OUT, stderr=subprocess.DEVNULL)
asm = open(OUT + "/trifinger_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
# function table of the source
starts, names = [], []
for n, line in enumerate(open(SRC), 1):
    m = re.match(r"\s*(?:template\s*<[^>]*>\s*)?(?:DEV|__global__)\s+[\w:<>\s\*&]*?\b(\w+)\s*\(", line)
    if m and not line.lstrip().startswith("//"):
        starts.append(n); names.append(m.group(1))
src_lines = open(SRC).read().split("\n")
def block_range(marker):
    """(first, last) source line of the brace block opened on the line containing `marker`"""
    for n, line in enumerate(src_lines, 1):
        if marker in line:
            depth = 0
            for mline in range(n, len(src_lines) + 1):
                depth += src_lines[mline - 1].count("{") - src_lines[mline - 1].count("}")
                if depth == 0 and mline > n or (depth == 0 and "{" in src_lines[mline - 1] and "}" in src_lines[mline - 1]):
                    return n, mline
    return 0, -1
PGS = block_range("for (int it = 0; it < P.iters; ++it) {")
NANG = block_range("if (!(acc == 0.0f)) {")
RESET = block_range("    if (rflag) {")
GOALR = block_range("    if (gflag) {")
def weight(lines, chain):
    """estimated executions per control step of an instruction with this inline chain (headline workload: 2 substeps,
    8 sweeps, resets in ~8 % of the waves)"""
    for ln, f in zip(lines, chain):
        if f == "substep":
            return 16.0 if PGS[0] <= ln <= PGS[1] else 2.0
        if f == "post_step_env" and NANG[0] <= ln <= NANG[1]:
            return 0.0
        if f == "apply_resets" and (RESET[0] <= ln <= RESET[1] or GOALR[0] <= ln <= GOALR[1]):
            return 0.08
    return 1.0
def func_of(line):
    i = bisect.bisect_right(starts, line) - 1
    return names[i] if i >= 0 else "?"
inside, cur, chain = False, None, ()
per_func, per_line, per_top = collections.Counter(), collections.Counter(), collections.Counter()
dyn_func, dyn_top, dyn_total, lines = collections.Counter(), collections.Counter(), 0.0, []
TOP = {"substep", "post_step_env", "k_step", "apply_resets"}
total = 0
for l in asm:
    if re.match(r"^%s[^ ]*:" % re.escape(K), l):
        inside = True; continue
    if inside and l.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    if re.match(r"\s*\.loc\s", l):
        lines = [int(x) for x in re.findall(r"trifinger_hip\.hip:(\d+)", l)]
        lines = [x for x in lines if x > 0]
        if lines:
            cur, chain = lines[0], tuple(func_of(x) for x in lines)
        continue
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_)", op):
        continue
    if FOCUS and FOCUS not in chain:
        continue
    total += 1
    per_func[func_of(cur) if cur else "?"] += 1
    per_line[cur] += 1
    # the function called directly from one of the top-level bodies
    top = "?"
    for idx, f in enumerate(chain):
        if f in TOP:
            top = f + " > " + (chain[idx - 1] if idx > 0 else "(own code)")
            break
    per_top[top] += 1
    wgt = weight(lines, chain)
    dyn_total += wgt
    dyn_func[func_of(cur) if cur else "?"] += wgt
    dyn_top[top] += wgt
print(f"{K}: {total} instructions (static)")
for f, c in per_func.most_common(40):
    print(f"  {f:28s} {c:6d}  {100.0 * c / total:5.1f} %")
print(f"estimated dynamic instructions per control step: {dyn_total:.0f}")
for f, c in dyn_top.most_common(30):
    print(f"  {f:50s} {c:8.0f}  {100.0 * c / dyn_total:5.1f} %")
print("dynamic, by innermost function:")
for f, c in dyn_func.most_common(30):
    print(f"  {f:50s} {c:8.0f}  {100.0 * c / dyn_total:5.1f} %")
print("by call site under the top-level bodies:")
for f, c in per_top.most_common(40):
    print(f"  {f:50s} {c:6d}  {100.0 * c / total:5.1f} %")
print("hottest source lines:")
src = open(SRC).read().split("\n")
for ln, c in per_line.most_common(int(os.environ.get('TOP_LINES', '25'))):
    print(f"  {c:5d}  L{ln}: {src[ln - 1].strip()[:110] if ln else ''}")
