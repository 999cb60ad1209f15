#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the fused TriFinger step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`:
one process per GPU, envs sharded with no data-path collective; RNG is keyed by global env id.  RCCL is used only for the
timing barrier and the max-over-ranks reduction of the elapsed time.  `value` is the WEAK-scaling figure (every rank steps
`--envs` environments: 65536 per GPU); beside it the same line carries the STRONG-scaling reading of the metric
("@65536 envs, 1/2/4/8 GPUs"): `value_strong_65536_total` = a second timed leg in which the 65536 envs of the headline are
partitioned over the ranks (shard_range: 65536 / N each, the step kernel's 256-register instantiation below 32768 per GPU).

A "step" is one control step (tf_step: masked resets, torque law, decimation x substeps of physics,
obs/states/rewards/termination) over the rank's whole batch, with synthetic random actions
2*U[0,1)-1 that are already resident in HBM (a ring of pre-generated action tensors).

Workload = BASELINE.json configs[2]: trifinger_difficulty_4, 65536 envs per GPU, torque mode, Hydra
defaults of scripts/rlg_hydra.py:58-118 + difficulty-4 reward schedule (:140-182), asymmetric obs on
(the shipped resources/config/rlg/asymm.yaml default), episode_length 750.

Rank 0 prints ONE JSON line with `roofline` (the fused step kernel vs the HBM roofline, as the north star asks;
what binds it is the serial chain of the contact solve; HBM traffic and issue counters are parsed
from the newest matching profiles/r*_pmc.txt, nothing is hard-coded) and `cpu_baseline` (this repo's CPU oracle on the
host cores - the reference's IsaacGym CPU pipeline cannot be run).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from leibnizgym_amd.engine import TrifingerEngine, make_config  # noqa: E402
from leibnizgym_amd import _capi  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3         # vector FP32 peak, for the secondary figure
N_SIMD = 256 * 4                 # SIMDs of the chip (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32)
VALU_CYCLES_PER_INST = 2.0       # a wave64 VALU instruction occupies a SIMD-32 for 2 cycles
TF_OBS_BASE, TF_STATES_EXTRA = 32, 72     # include/trifinger.h: obs = 32 + A values, states = obs + 72
BYTES_PER_ENV_STEP = {False: 623, True: 1075}     # symmetric / asymmetric obs (algorithmic)
ACTION_READ_BYTES = 36                            # ... of which the [N, 9] action read, which the instantiation with the fused action source does not perform
# exact fp32 operation count of one env-step of this workload on a scalar machine (the oracle's instrumented build, oracle/tf_flops.h), default model of
# API 8 (the six middle-distal finger-finger pairs included: their geometry runs for every env): 2722 add + 5039 mul + 18 div + 12 sqrt + 7390 fma counted
# twice; beside them 3995 comparisons / min / max / abs and 116 conversions (the fast contact set of API <= 7: 19.7 kFLOP, profiles/r3_l_flops.txt).
# tests/test_flop_count.py holds this constant to the count.  It replaces SURVEY.md 8(d)'s 33 kFLOP paper estimate.  The GPU executes more than this
# (a wavefront runs a contact row whenever one of its 64 envs needs it): that is valu_lane_ops_per_env_step, from the counters.
FLOPS_PER_ENV_STEP = 22.6e3


def load_pmc_profile(n, asym, ext=False, variant="narrow"):
    """Per-launch counters of the fused step kernel from the newest profiles/r*_pmc.txt whose header names this workload
    (written by tools/profile_round.sh; rocprofv3 --pmc passes, raw counter expressions).  Returns (dict, path) or
    (None, None): nothing is hard-coded here, a profile of another N / kernel is not used."""
    import glob
    import re
    kname = kernel_name(asym, 9, ext, True, variant)
    want = f"# workload: N={n} asym={asym} kernel={kname}"
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc.txt")), reverse=True):
        text = open(path).read()
        if want not in text:
            continue
        vals = {}
        for line in text.splitlines():
            if line.startswith(kname):
                tok = line.split()                     # ... <counter> <n> <mean> <min> <max>; the kernel name may be cut short
                if len(tok) >= 6 and re.fullmatch(r"[A-Z_0-9a-z]+", tok[-5]) and tok[-4].isdigit():
                    vals[tok[-5]] = float(tok[-3])
        if vals:
            return vals, os.path.relpath(path, REPO)
    return None, None


def load_pmc_calibration():
    """Correction factors (true bytes / counter) for FETCH_SIZE / WRITE_SIZE in the access patterns of the fused step, from the
    newest profiles/r*_pmc_calibration.txt (tools/pmc_calibrate.sh: kernels with known byte counts under the same two rocprofv3
    passes; MI355X_MICROARCH.md asks for exactly this where an access pattern is not the calibrated 16-B/lane stream)."""
    import glob
    KIB_ROWS, KIB_TILE = 64 * 65536 * 4 / 1024.0, 65536 * 113 * 4 / 1024.0
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_calibration.txt")), reverse=True):
        got = {}
        for line in open(path):
            tok = line.split()
            if line.startswith("cal_") and len(tok) >= 6 and tok[-5] in ("FETCH_SIZE", "WRITE_SIZE"):
                got[(tok[0].split("(")[0], tok[-5])] = float(tok[-3])
        try:
            return {"fetch_rows_read_only": KIB_ROWS / got[("cal_read_rows", "FETCH_SIZE")],       # rows only read: the counter shows half
                    "fetch_rows_in_place": KIB_ROWS / got[("cal_rw_rows", "FETCH_SIZE")],         # rows read and rewritten in place
                    "write_rows_in_place": KIB_ROWS / got[("cal_rw_rows", "WRITE_SIZE")],         # ... their stores are counted twice
                    "write_rows_fresh": KIB_ROWS / got[("cal_write_rows", "WRITE_SIZE")],
                    "write_tile": KIB_TILE / got[("cal_write_tile", "WRITE_SIZE")]}, os.path.relpath(path, REPO)
        except KeyError:
            continue
    return None, None


D4_REWARDS = {                   # scripts/rlg_hydra.py:140-174
    "finger_move_penalty": {"activate": True, "weight": -0.1},
    "finger_reach_object_rate": {"activate": True, "norm_p": 2, "weight": -250,
                                 "thresh_sched_start": 0, "thresh_sched_end": 1e7},
    "object_dist": {"activate": True, "weight": 2000, "thresh_sched_start": 0, "thresh_sched_end": 10e10},
    "object_rot": {"activate": True, "weight": 2000, "epsilon": 0.01, "scale": 3.0,
                   "thresh_sched_start": 1e7, "thresh_sched_end": 1e10},
    "object_rot_delta": {"activate": False, "weight": -250},
    "object_move": {"activate": False, "weight": -750},
}
D4_SUCCESS = {"activate": False, "bonus": 5000.0, "orientation_tolerance": 0.25, "position_tolerance": 0.02}


D1_REWARDS = {                   # scripts/rlg_hydra.py:83-109
    "finger_move_penalty": {"activate": True, "weight": -0.1},
    "finger_reach_object_rate": {"activate": True, "norm_p": 2, "weight": -750},
    "object_dist": {"activate": True, "weight": 2000},
    "object_rot": {"activate": False, "weight": 300},
    "object_rot_delta": {"activate": False, "weight": -250},
    "object_move": {"activate": False, "weight": -750},
}
# EVERY domain-randomisation feature of the build (BASELINE configs[3] "full domain randomisation"): the six scale factors,
# observation noise, action repeat AND the rest of the reference's intent list (trifinger_env.py:385-393): robot base / stage
# position offsets and per-body friction - the latter select the EXT kernel instantiation k_env<..., true>.
FULL_DR = {"activate": True, "cube_mass": (0.7, 1.3), "cube_size": (0.9, 1.1), "friction": (0.7, 1.3),
           "motor_torque": (0.9, 1.1), "link_mass": (0.9, 1.1), "restitution": (0.5, 1.5), "obs_noise": 0.02,
           "action_repeat_prob": 0.1, "robot_base_position": (0.005, 0.005, 0.003), "stage_position": (0.005, 0.005),
           "friction_robot": (0.8, 1.2), "friction_object": (0.8, 1.2), "friction_stage": (0.8, 1.2)}


def kernel_name(asym, action_dim=9, ext=False, fused_actions=True, variant="narrow"):
    """rocprofv3's name of the fused-step instantiation a workload launches (EXT: extended DR or the box object; MODE 127 = the step
    with the action source fused in - tf_step_random, what `value` times -, 63 = the step that reads a resident action tensor; the last
    two arguments: the 256-register instantiation tf_create picks up to 32768 envs per handle, and its form with helper wavefronts - up to
    16384 envs; `variant` is TrifingerEngine.kernel_variant)."""
    b = lambda x: "true" if x else "false"        # noqa: E731
    return f"k_env<{action_dim}, false, {b(asym)}, {127 if fused_actions else 63}, {int(ext)}, {b(variant != 'narrow')}, {b(variant == 'wide_helpers')}>"


def workload_kwargs(asym, difficulty=4, dr=False):
    kw = dict(command_mode="torque", task_difficulty=difficulty, asymmetric_obs=asym, normalize_action=True,
              normalize_obs=True, apply_safety_damping=True, episode_length=750, control_decimation=1,
              robot_reset="default", object_reset="random", reward_terms=D4_REWARDS if difficulty == 4 else D1_REWARDS,
              success=D4_SUCCESS, dt=0.02, substeps=2, solver_iterations=8)
    if dr:
        kw["domain_randomization"] = FULL_DR
    return kw


def cpu_baseline(asym, budget_s=12.0):
    """Time the CPU oracle (same algorithm, scalar C, OpenMP over envs) on a bounded sample of the workload,
    on one thread and on every host core."""
    import ctypes
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "-s"], stdout=subprocess.DEVNULL)
    so = os.path.join(REPO, "oracle", "_build", "libtrifinger_oracle_omp.so")
    lib = _capi.TfLib(so)
    lib.dll.tfo_omp_threads.argtypes = [ctypes.c_int]
    lib.dll.tfo_omp_threads.restype = ctypes.c_int
    def usable_cpus():
        # what this process may actually use: affinity mask, further limited by a cgroup CPU quota if there is one
        n_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                n_aff = max(1, min(n_aff, int(float(quota) / float(period) + 0.5)))
        except Exception:
            pass
        return n_aff
    cores = usable_cpus()
    n = 65536                     # the headline batch
    acts = [(torch.rand(n, 9) * 2 - 1).contiguous() for _ in range(4)]

    def run(threads, budget):
        used = lib.dll.tfo_omp_threads(threads)
        eng = TrifingerEngine(make_config(lib, n, seed=7, **workload_kwargs(asym)), device="cpu", lib=lib)
        eng.reset()
        eng.step(acts[0])                                   # warm-up
        t0 = time.perf_counter()
        eng.step(acts[1])
        one = time.perf_counter() - t0
        steps = max(1, min(1000, int(budget / max(one, 1e-6))))
        t0 = time.perf_counter()
        for k in range(steps):
            eng.step(acts[k % 4])
        el = time.perf_counter() - t0
        eng.close()
        return used, steps, el, n * steps / el

    _, s1, e1, v1 = run(1, budget_s * 0.3)
    # the container may expose more logical CPUs than it is allowed to run on: try a few team sizes, keep the best
    best = None
    for threads in sorted({min(cores, 8), min(cores, 32), cores}):
        r = run(threads, budget_s * 0.25)
        if best is None or r[3] > best[3]:
            best = r
    used, sa, ea, va = best
    # the per-GPU size of BASELINE configs[1] / [3] / [4] as well: same team size, a small budget
    n_small = 8192
    acts_s = [(torch.rand(n_small, 9) * 2 - 1).contiguous() for _ in range(4)]

    def run_small(threads, budget):
        lib.dll.tfo_omp_threads(threads)
        eng = TrifingerEngine(make_config(lib, n_small, seed=7, **workload_kwargs(asym)), device="cpu", lib=lib)
        eng.reset()
        eng.step(acts_s[0])
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < budget:
            eng.step(acts_s[k % 4]); k += 1
        el = time.perf_counter() - t0
        eng.close()
        return n_small * k / el
    v_small = run_small(used, 1.5)
    # beside the bit-exact oracle: the same source built for THIS host (-O3 -march=native, contraction allowed; oracle/Makefile `fast`): not a
    # checker, only the CPU's best effort at the same algorithm
    v_fast = None
    try:
        subprocess.check_call(["make", "-B", "-C", os.path.join(REPO, "oracle"), "-s", "fast"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)      # -B: -march=native is THIS host's
        bitexact = lib
        lib = _capi.TfLib(os.path.join(REPO, "oracle", "_build", "libtrifinger_oracle_fast.so"))
        lib.dll.tfo_omp_threads.argtypes = [ctypes.c_int]
        lib.dll.tfo_omp_threads.restype = ctypes.c_int
        v_fast = run(used, budget_s * 0.2)[3]
        lib = bitexact
    except Exception:
        v_fast = None
    return {"value": va, "unit": "env-steps/s", "cores": used, "kind": "port",
            "single_thread_value": v1, "value_at_8192_envs": v_small,
            "best_effort_value": v_fast,
            "best_effort_is": "the same source compiled -O3 -march=native -ffp-contract=fast on this host (not bit-identical to the product: timed only)",
            "sample": f"{n} envs x {sa} steps of the same workload on {used} OpenMP threads (static schedule over envs, "
                      f"{ea:.1f} s) and x {s1} steps on 1 thread ({e1:.1f} s); reference IsaacGym CPU pipeline not "
                      f"available, baseline is this repo's CPU oracle (oracle/tf_oracle.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--symmetric", action="store_true", help="asymmetric_obs=False (obs only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--difficulty", type=int, default=4, choices=(1, 4),
                    help="secondary runs only (BASELINE configs[1]: --difficulty 1 --envs 8192); the headline is 4")
    ap.add_argument("--dr", action="store_true",
                    help="secondary runs only: every domain-randomisation feature on (BASELINE configs[3]: --dr --envs 16384)")
    ap.add_argument("--box", action="store_true",
                    help="secondary runs only: the phase-3 cuboid (20 x 80 x 20 mm, density 500) instead of the cube - the EXT "
                         "kernel instantiation with the inertia-scaled solve")
    ap.add_argument("--stats-every", type=int, default=0,
                    help="all-reduce the episode statistics over the ranks every K steps on a side stream (the optional "
                         "exchange of the north star; RCCL on a multi-GPU run); 0: off")
    ap.add_argument("--settle", type=int, default=-1,
                    help="untimed steps of the same workload BEFORE the W warm-up steps that put the env population into the steady "
                         "state of a long run: the per-env step counters are spread uniformly over the episode length first (an env "
                         "population that has been stepping for a long time is at every episode phase at once), then this many steps are "
                         "taken.  Default: one episode length (750), so that every env has gone through a time-out reset.  0: none - the timed "
                         "region then starts --warmup steps after a reset of ALL envs (a correlated transient: "
                         "profiles/r4_a_driver_repro.txt)")
    ap.add_argument("--no-fast-contact-leg", action="store_true",
                    help="skip the second timed leg on the contact set of API <= 7 (`value_fast_contact_set`)")
    ap.add_argument("--strong-total", type=int, default=-1,
                    help="global env count of the strong-scaling leg (partitioned over the ranks); default: --envs (65536), 0: no such leg")
    ap.add_argument("--time-window", type=int, default=8,
                    help="one HIP event pair per window of W consecutive k_step launches of the timed region (an event "
                         "pair costs ~3 us of stream time: per launch it would slow the region it measures and read "
                         "2-3 us long; over 8 back-to-back launches it is amortised); 0: no kernel timing")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # Test hook for boxes with ONE GPU: every rank uses cuda:0 and the (timing-only) collectives go through gloo, so that the
    # multi-process path - rank offsets, barrier, max over ranks, rank-0 JSON - can be exercised without a second device.
    one_device = os.environ.get("TF_BENCH_SINGLE_DEVICE_TEST", "") == "1"
    if one_device:
        local_rank = 0
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if world != max(1, args.gpus):
        sys.exit(f"bench.py --gpus {args.gpus} needs {args.gpus} ranks (WORLD_SIZE={world}): launch it as "
                 f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                 f"bench.py --gpus {args.gpus} ...`")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path in the product)"
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    asym = not args.symmetric
    n = args.envs
    headline = args.difficulty == 4 and not args.dr and not args.box
    lib = _capi.load_hip_library()
    cfg = make_config(lib, n, seed=7, env_id_offset=rank * n, global_num_envs=world * n,
                      model=lib.box_model((0.02, 0.08, 0.02), 500.0) if args.box else None,
                      **workload_kwargs(asym, args.difficulty, args.dr))
    eng = TrifingerEngine(cfg, device=dev, lib=lib)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    ring = [(torch.rand(n, eng.action_dim, device=dev, generator=gen) * 2 - 1).contiguous() for _ in range(16)]
    eng.reset()
    # ---- steady state of the workload (untimed set-up, like generating the synthetic inputs) ----
    # A reset of ALL envs at once is a state a long run never is in: every finger starts from the same pose, the contacts of all
    # 65536 envs arrive in the same frames (frames 16-21 after the reset cost 75-77 us against 71 us in the steady state:
    # profiles/r4_a_driver_repro.txt) and all episodes would time out in the same step.  The benchmark measures the steady state
    # BASELINE.md section 4 asks for (time-out resets and both reward-schedule regimes inside the window): the step counters are
    # spread over the episode, then one episode length of steps is taken, so every env has been through a time-out reset and the
    # population sits at every episode phase at once.  None of this is timed; --settle 0 switches it off.
    ep_len = int(cfg.episode_length)
    settle = args.settle if args.settle >= 0 else ep_len
    if settle > 0 and ep_len > 0:
        eng.steps.copy_(torch.randint(0, ep_len, (n,), device=dev, generator=gen, dtype=torch.int64))
        for k in range(settle):
            eng.step_random()
    for k in range(args.warmup):
        eng.step_random()
    # BASELINE.md section 4 asks for both reward-schedule regimes inside the timed window: the finger_reach term switches off when env_steps_count =
    # frame count x global envs passes 1e7 (frame 153 at 65536 envs), long before the prelude ends.  The frame counter (reward schedule, keys of the
    # action / noise draws; nothing else) is therefore put back so that the switch falls into the middle of the timed steps.
    switch_frame = int(-(-1e7 // (world * n)))
    eng.frame_count = max(0, switch_frame - args.steps // 2)
    frame_first = eng.frame_count

    def barrier():
        if distributed:
            dist.barrier()
        # the stream is drained by polling an event first: a blocking wait wakes the host 20-60 us late, which is 2-4 % of the
        # 1.4 ms a `--steps 20` window lasts (profiles/r4_b_driver_repro.txt: same kernel time, value 8.6 ... 9.2e8)
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize()

    reducer = None
    if args.stats_every > 0:
        from leibnizgym_amd.sharding import EpisodeStatsReducer
        if not distributed:           # single rank: a world of one, so that the same code path runs
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="gloo" if one_device else "nccl", rank=0, world_size=1)
        reducer = EpisodeStatsReducer(eng, world * n, every=args.stats_every)
    # ---- the timed region: EXACTLY --steps steps, actions 2*U-1 generated on the device in every step by the
    # step itself (tf_step_random: Philox draws inside the launch) ----
    eng.enable_kernel_timing(8192 if args.time_window > 0 else 0, max(1, args.time_window))
    # the driver's window is 20 steps = 1.4 ms: a cyclic-GC pass of the interpreter inside it (torch is loaded: a full collection takes milliseconds) would
    # be most of the measurement - collect now, keep the collector off while the timed regions run (the loop itself allocates nothing that needs it)
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        eng.step_random()
        if reducer is not None:
            reducer.step()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = eng.kernel_time_ms()
    global_stats = reducer.result().cpu().tolist() if reducer is not None else None
    local_stats = eng.info.cpu().tolist() if reducer is not None else None
    eng.enable_kernel_timing(0)
    # beside it, over the same number of steps: the step fed from a ring of 16 RESIDENT action tensors (what rounds 1-3 printed as
    # `value`) ...
    barrier()
    t2 = time.perf_counter()
    for k in range(args.steps):
        eng.step(ring[k % len(ring)])
    barrier()
    elapsed_ring = time.perf_counter() - t2
    # ... and with torch.rand(N, A)*2-1 generated inside the loop (three extra elementwise launches per step), a quarter of the steps
    gen_steps = max(1, args.steps // 4)
    barrier()
    t1 = time.perf_counter()
    for k in range(gen_steps):
        eng.step(torch.rand(n, eng.action_dim, device=dev, generator=gen) * 2 - 1)
    barrier()
    elapsed_gen = time.perf_counter() - t1
    if distributed:
        t = torch.tensor([elapsed, elapsed_gen, elapsed_ring], device=dev if not one_device else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_gen, elapsed_ring = float(t[0].item()), float(t[1].item()), float(t[2].item())

    # sanity on what was just timed: finite state, resets happened if steps crossed an episode boundary
    assert torch.isfinite(eng.state).all(), "non-finite state after the timed region"

    # ---- the STRONG-scaling reading of the metric ("@65536 envs, 1/2/4/8 GPUs"; SURVEY 8e partitions N over G): the same workload with
    # --strong-total envs IN TOTAL, rank r stepping shard_range(total, r, world) of them (global env ids and the reward schedule as in one
    # 65536-env engine: the union of the shards is bit-identical to it, tests/test_multi_rank.py).  Same prelude, same warm-up, same number
    # of timed steps, same barriers and max over ranks.  At one rank it is the weak leg itself and is not run twice.
    strong_total = args.strong_total if args.strong_total >= 0 else n
    strong = None
    if strong_total > 0 and (world > 1 or strong_total != n):
        from leibnizgym_amd.sharding import shard_range
        off, cnt = shard_range(strong_total, rank, world)
        cfg_s = make_config(lib, cnt, seed=7, env_id_offset=off, global_num_envs=strong_total,
                            model=lib.box_model((0.02, 0.08, 0.02), 500.0) if args.box else None,
                            **workload_kwargs(asym, args.difficulty, args.dr))
        eng_s = TrifingerEngine(cfg_s, device=dev, lib=lib)
        eng_s.reset()
        if settle > 0 and ep_len > 0:
            eng_s.steps.copy_(torch.randint(0, ep_len, (cnt,), device=dev, generator=gen, dtype=torch.int64))
            for k in range(settle):
                eng_s.step_random()
        for k in range(args.warmup):
            eng_s.step_random()
        eng_s.frame_count = frame_first              # the same reward-schedule window as the weak leg (its frame counter was set back too)
        eng_s.enable_kernel_timing(8192 if args.time_window > 0 else 0, max(1, args.time_window))
        barrier()
        t3 = time.perf_counter()
        for k in range(args.steps):
            eng_s.step_random()
        barrier()
        elapsed_s = time.perf_counter() - t3
        ks_ms, ks_n = eng_s.kernel_time_ms()
        assert torch.isfinite(eng_s.state).all(), "non-finite state after the strong-scaling leg"
        if distributed:
            t = torch.tensor([elapsed_s], device=dev if not one_device else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_s = float(t[0].item())
        strong = {"global_envs": strong_total, "envs_per_gpu": cnt, "ms_per_step": elapsed_s / args.steps * 1e3,
                  "value": strong_total * args.steps / elapsed_s, "kernel_variant": eng_s.kernel_variant,
                  "kernel": kernel_name(asym, eng_s.action_dim, 2 if args.box else (1 if args.dr else 0), True, eng_s.kernel_variant),
                  "kernel_avg_us_rank0": (ks_ms / max(ks_n, 1)) * 1e3}
        eng_s.close()

    # ---- beside `value`: the same workload on the contact set of API <= 7.  Since API 8 the default model holds the reference's self-collision set
    # (TfModel.ff_middle_pairs = 1: the middle link of a finger against the fingertip of another, trifinger_env.py:811-812) and `value` is timed on
    # it; `native.ff_middle_pairs: false` restores the distal pairs only - the faster step every earlier round's `value` was.  Same prelude, same
    # warm-up, same frame window, same number of timed steps, same barriers and max over ranks.
    fast = None
    if not args.no_fast_contact_leg:
        model_f = lib.box_model((0.02, 0.08, 0.02), 500.0) if args.box else lib.default_model()
        model_f.ff_middle_pairs = 0
        cfg_f = make_config(lib, n, seed=7, env_id_offset=rank * n, global_num_envs=world * n, model=model_f, **workload_kwargs(asym, args.difficulty, args.dr))
        eng_f = TrifingerEngine(cfg_f, device=dev, lib=lib)
        eng_f.reset()
        if settle > 0 and ep_len > 0:
            eng_f.steps.copy_(torch.randint(0, ep_len, (n,), device=dev, generator=gen, dtype=torch.int64))
            for k in range(settle):
                eng_f.step_random()
        for k in range(args.warmup):
            eng_f.step_random()
        eng_f.frame_count = frame_first
        eng_f.enable_kernel_timing(8192 if args.time_window > 0 else 0, max(1, args.time_window))
        barrier()
        t4 = time.perf_counter()
        for k in range(args.steps):
            eng_f.step_random()
        barrier()
        elapsed_f = time.perf_counter() - t4
        kf_ms, kf_n = eng_f.kernel_time_ms()
        assert torch.isfinite(eng_f.state).all(), "non-finite state after the fast-contact-set leg"
        if distributed:
            t = torch.tensor([elapsed_f], device=dev if not one_device else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_f = float(t[0].item())
        fast = {"value": world * n * args.steps / elapsed_f, "ms_per_step": elapsed_f / args.steps * 1e3, "kernel_avg_us_rank0": (kf_ms / max(kf_n, 1)) * 1e3,
                "contact_set": "TfModel.ff_middle_pairs = 0 (`native.ff_middle_pairs: false`): finger-finger contacts between the three distal pairs only"}
        eng_f.close()

    total_env_steps = world * n * args.steps
    value = total_env_steps / elapsed
    kern_avg_s = (kern_ms / max(kern_n, 1)) * 1e-3
    # `value` launches the step with the action source fused in (tf_step_random): it draws its actions and never reads the action tensor, so the
    # 36 B/env of SURVEY's figure that are the action read are NOT counted for it (the resident-action instantiation would count them)
    alg_bytes = BYTES_PER_ENV_STEP[asym] - ACTION_READ_BYTES
    bytes_per_launch = alg_bytes * n
    achieved_gbs = bytes_per_launch / kern_avg_s / 1e9 if kern_n else 0.0
    ext = 2 if args.box else (1 if args.dr else 0)     # extended DR -> EXT = 1, box object -> EXT = 2 instantiation of the fused step
    pmc, pmc_path = load_pmc_profile(n, asym, ext, eng.kernel_variant) if (headline or ext) else (None, None)      # counters of the kernel `value` launches
    traffic = traffic_raw = issue = None
    traffic_how = None
    if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        traffic_raw = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0      # KiB per dispatch -> bytes
        cal, cal_path = load_pmc_calibration()
        if cal:
            # write-only API tensors of a launch (obs, states, action_buf as tiles; 1.0 in the calibration) vs state rows that the
            # step reads and rewrites in place (their stores are tallied twice, their loads once)
            tile = n * 4.0 * (TF_OBS_BASE + eng.action_dim + ((TF_OBS_BASE + eng.action_dim + TF_STATES_EXTRA) if asym else 0) + eng.action_dim)
            w_raw = pmc["WRITE_SIZE"] * 1024.0
            w_true = tile * cal["write_tile"] + max(0.0, w_raw - tile) * cal["write_rows_in_place"]
            traffic = pmc["FETCH_SIZE"] * 1024.0 * cal["fetch_rows_in_place"] + w_true
            traffic_how = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {pmc_path}, corrected with the factors measured on "
                           f"known byte counts in this kernel's access patterns ({cal_path}): stores of rows rewritten in place x{cal['write_rows_in_place']:.3f} "
                           f"(the counter tallies them twice), tile stores x{cal['write_tile']:.3f} ({tile / 1e6:.1f} MB per launch: obs, states, action_buf), "
                           f"loads of those rows x{cal['fetch_rows_in_place']:.3f}; the read-only rows (action 36 B, goal 40 B per env) may be "
                           f"under-counted by up to half (x{cal['fetch_rows_read_only']:.2f} for a pure read stream); traffic_raw = the two counters as printed")
        else:
            traffic = traffic_raw
            traffic_how = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, raw counters), {pmc_path}"
    if pmc and "SQ_INSTS_VALU" in pmc and "SQ_WAVE_CYCLES" in pmc:
        issue = {
            "valu_insts_per_launch": pmc["SQ_INSTS_VALU"], "waves_per_launch": pmc.get("SQ_WAVES"),
            "wave_cycles_per_launch": pmc["SQ_WAVE_CYCLES"] * 4.0,              # the counter is in quad-cycles
            # fraction of the resident wave-cycles in which a VALU instruction of that wave occupies the SIMD
            "valu_issue_frac": pmc["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / (pmc["SQ_WAVE_CYCLES"] * 4.0),
            "wait_frac": pmc.get("SQ_WAIT_ANY", 0.0) / pmc["SQ_WAVE_CYCLES"],
            "source": f"rocprofv3 --pmc SQ_*, {pmc_path}",
        }
    clock_ghz = 2.4
    simd_busy = (pmc["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / (N_SIMD * kern_avg_s * clock_ghz * 1e9)) if (pmc and kern_n and "SQ_INSTS_VALU" in pmc) else None
    secondary = "" if headline else " [secondary run: not the headline workload]"
    out = {
        "metric": "env-steps/sec (whole node), trifinger_difficulty_4 @65536 envs, 1/2/4/8 GPUs" + secondary,
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        # the strong-scaling reading of the metric: 65536 envs IN TOTAL partitioned over the ranks (at one rank: the weak figure itself)
        "value_strong_65536_total": (strong["value"] if strong else (value if strong_total == n else None)) if strong_total == 65536 else None,
        "strong_scaling": strong if strong else ({"global_envs": n, "envs_per_gpu": n, "ms_per_step": elapsed / args.steps * 1e3, "value": value,
                                                  "same_as": "value (one rank: the weak and the strong partition coincide)"} if strong_total == n else None),
        # the default model = the reference's contact set since API 8; the opt-out (the step of every earlier round) beside it
        "contact_set": "TfModel.ff_middle_pairs = 1 (default since API 8): every finger-finger pair the reference's self-collision group holds among the "
                       "modelled shapes - the three distal pairs and the six middle-distal pairs (trifinger_env.py:811-812)",
        "value_fast_contact_set": fast["value"] if fast else None,
        "fast_contact_set": fast,
        "comparability": "`value` of rounds 1-5 was timed on the fast contact set (compare their `value` with `value_fast_contact_set`); rounds 1-4 "
                         "did not set the frame counter back after the prelude (reward-schedule switch outside their window)",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "value_resident_actions": world * n * args.steps / elapsed_ring,
        "value_with_torch_action_generation": world * n * gen_steps / elapsed_gen,
        "action_generation": f"`value`: every step generates its actions 2*U-1 on the device (BASELINE.md section 4) inside the launch itself "
                             f"(tf_step_random: Philox draws keyed by global env id and frame count); `value_resident_actions`: the same "
                             f"{args.steps} steps fed from a ring of 16 resident action tensors (what rounds 1-3 reported as `value`); "
                             f"`value_with_torch_action_generation`: {gen_steps} steps with torch.rand(N, A)*2-1 generated inside the loop "
                             f"(three extra elementwise launches per step)",
        "reward_schedule_in_window": (f"frames {frame_first}..{frame_first + args.steps} are timed (the frame counter is set back after the prelude): the "
                                      f"finger_reach_object_rate term switches off at frame {switch_frame} (env_steps_count 1e7), "
                                      + ("inside the window" if frame_first < switch_frame <= frame_first + args.steps else "outside the window")),
        "steady_state_prelude": (f"untimed set-up before the {args.warmup} warm-up steps: per-env step counters spread uniformly over the episode "
                                 f"length ({ep_len}), then {settle} steps of the same workload, so that the env population is at every episode "
                                 f"phase at once as in a long run (a timed region right after a reset of all envs is a correlated transient: "
                                 f"profiles/r4_a_driver_repro.txt); --settle 0 switches it off") if (settle > 0 and ep_len > 0) else "none (--settle 0)",
        "config": {
            "workload": f"trifinger_difficulty_{args.difficulty}{' + full domain randomisation' if args.dr else ''}{' + phase-3 cuboid object' if args.box else ''}, "
                        f"{n} envs/GPU x {world} GPU, torque mode, random actions 2*U-1, "
                        f"asymmetric_obs={asym}, episode_length 750, dt 0.02, 2 substeps, 8 solver iterations, "
                        f"control_decimation 1 (BASELINE.json configs[{2 if headline else (3 if args.dr else 1)}]{'' if not args.box else ' with the object swapped'})",
            "envs_per_gpu": n,
            "global_envs": world * n,
            "strong_scaling_workload": (f"the same workload with {strong_total} envs in total, {strong_total // world} per GPU x {world} GPU "
                                        f"(`value_strong_65536_total` / `strong_scaling`)") if strong_total > 0 else None,
            "asymmetric_obs": asym,
            "parallelism": f"env-shard x{world} (no data-path collective"
                           + (f"; episode statistics all-reduced every {args.stats_every} steps on a side stream)" if reducer else ")"),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbs,
            "achieved_is": "algorithmic bytes per launch (SURVEY 8d: 1075 B/env-step asymmetric, 623 B symmetric, MINUS the 36 B action read, which the "
                           "timed instantiation - actions drawn inside the launch - does not perform; the solver's warm-start rows are an "
                           "implementation choice and are not counted) / measured kernel time",
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_raw": traffic_raw,
            "traffic_source": traffic_how,
            "kernel": kernel_name(asym, eng.action_dim, ext, True, eng.kernel_variant),
            "kernel_variant": eng.kernel_variant,
            "kernel_avg_us": kern_avg_s * 1e6,
            "kernel_launches_timed": kern_n,
            "kernel_timing": f"one HIP event pair on the launch stream around every window of {max(1, args.time_window)} "
                             f"consecutive launches of the fused step kernel in the timed region; kernel_avg_us = window time / launches",
            "algorithmic_bytes_per_env_step": alg_bytes,
            "algorithmic_bytes_per_env_step_survey": BYTES_PER_ENV_STEP[asym],
            "note": "north star asks for the HBM fraction; what binds the fused step is the serial dependency chain of the "
                    "Gauss-Seidel solve (one chain per 64 envs, carried by the cube wavefront of each workgroup) and "
                    "instruction issue, not bandwidth: see valu_issue and DESIGN.md section 4",
            "valu_issue": issue,
            "valu_issue_frac": issue["valu_issue_frac"] if issue else None,
            "simd_valu_busy": simd_busy,
            # measured operation count: vector-ALU lane-operations per env-step = SQ_INSTS_VALU x 64 lanes / N (every wave64 VALU
            # instruction of the launch, fp32 arithmetic and the integer / select / move instructions around it alike; an FMA is ONE
            # lane-operation).  The chip retires 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-operations/s (its 157.3 TFLOP/s
            # count an FMA as two), so valu_lane_ops_frac = lane-operations/s / 78.6e12.
            "valu_lane_ops_per_env_step": (pmc["SQ_INSTS_VALU"] * 64.0 / n) if (pmc and "SQ_INSTS_VALU" in pmc) else None,
            "valu_lane_ops_frac": (pmc["SQ_INSTS_VALU"] * 64.0 / kern_avg_s / (FP32_PEAK_TFLOPS * 0.5e12)) if (pmc and kern_n and "SQ_INSTS_VALU" in pmc) else None,
            "flops_per_env_step": FLOPS_PER_ENV_STEP,
            "flops_source": "exact count of the oracle's instrumented build on the default model of API 8 (tests/test_flop_count.py)",
            "fp32_frac": (FLOPS_PER_ENV_STEP * n / kern_avg_s / 1e12 / FP32_PEAK_TFLOPS) if kern_n else 0.0,
        },
    }
    if global_stats is not None:
        out["episode_stats_all_reduced"] = global_stats[:11]
        out["episode_stats_rank0"] = local_stats[:11]          # this rank's own statistics of the same step (equal to the reduced ones in a world of one)
    gc.enable()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(asym)
        print(json.dumps(out), flush=True)
    eng.close()
    if distributed or reducer is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
