"""Thin owner of the device buffers behind one native handle.

`TrifingerEngine` is the Python face of the inner boundary (include/trifinger.h): it allocates the
torch tensors the native library works on (ownership is inverted with respect to
`gymtorch.wrap_tensor`, reference trifinger_env.py:594-617), binds their addresses once, and
forwards `step`/`reset` to the fused HIP launch on torch's current stream.  No arithmetic of the
hot path lives here and there is no fallback: without the HIP library (or an injected one in the
tests) construction fails.
"""
import ctypes as C

import torch

C_NORM_INF = -1          # TF_NORM_INF of include/trifinger.h

from . import _capi as capi


def default_reward_terms():
    """Reward configuration of the env's default dict (reference trifinger_env.py:76-104)."""
    return {
        "finger_reach_object_rate": {"activate": True, "weight": -750, "norm_p": 2},
        "finger_move_penalty": {"activate": True, "weight": -0.1},
        "object_dist": {"activate": True, "weight": 2000},
        "object_rot": {"activate": True, "weight": 300},
        "object_rot_delta": {"activate": True, "weight": -250},
        "object_move": {"activate": True, "weight": -750},
    }


# defaults of each term's constructor (reference rewards.py:40-47,70,105-115,155-163,193-201,241-243)
_TERM_DEFAULT_WEIGHT = {
    "finger_reach_object_rate": -250, "finger_move_penalty": -1.0e-4, "object_dist": 2000,
    "object_rot": 100, "object_rot_delta": 100, "object_move": -750,
}


def fill_reward_terms(cfg, reward_terms):
    """Copy a reference-style `reward_terms` dict into TfConfig (names: rewards.py:267-274)."""
    for k, name in enumerate(capi.REWARD_TERM_ORDER):
        kw = dict(reward_terms.get(name, {"activate": False}))
        term = cfg.reward[k]
        term.activate = int(bool(kw.get("activate", False)))
        term.weight = float(kw.get("weight", _TERM_DEFAULT_WEIGHT[name]))
        if name == "object_rot_delta":
            term.sched_start = float(kw.get("linear_schedule_start", 0))
            term.sched_end = float(kw.get("linear_schedule_end", 0))
        elif name in ("finger_reach_object_rate", "object_dist", "object_rot"):
            term.sched_start = float(kw.get("thresh_sched_start", 0))
            term.sched_end = float(kw.get("thresh_sched_end", 0))
        else:
            term.sched_start = 0.0
            term.sched_end = 0.0
    norm_p = reward_terms.get("finger_reach_object_rate", {}).get("norm_p", 2)      # rewards.py:203: any p torch.norm takes
    if isinstance(norm_p, str):
        norm_p = float(norm_p)
    if norm_p == float("inf"):
        cfg.finger_reach_norm_p = C_NORM_INF
    elif float(norm_p) == int(norm_p) and 1 <= int(norm_p) <= 16:
        cfg.finger_reach_norm_p = int(norm_p)
    else:
        raise ValueError(f"finger_reach_object_rate.norm_p = {norm_p!r}: the native step builds integer p in 1..16 and inf")
    cfg.object_rot_scale = float(reward_terms.get("object_rot", {}).get("scale", 1.0))


def make_config(lib, num_envs, *, seed=0, env_id_offset=0, global_num_envs=0, command_mode="position",
                normalize_action=True, normalize_obs=True, apply_safety_damping=True, asymmetric_obs=False,
                enable_ft_sensors=False, task_difficulty=1, episode_length=750, control_decimation=1,
                robot_reset="default", dof_pos_stddev=0.4, dof_vel_stddev=0.2, object_reset="random",
                goal_rotation=False, goal_rotation_rate=0.5, reward_terms=None, success=None,
                dt=0.02, substeps=2, solver_iterations=8, solver_inner=1, gravity=(0.0, 0.0, -9.81), model=None,
                domain_randomization=None):
    """Build a TfConfig.  String options are validated here with the reference's ValueErrors."""
    if command_mode not in capi.COMMAND_MODES:
        raise ValueError(f"Invalid command mode. Input: {command_mode} not in ['torque', 'position'].")
    if robot_reset not in capi.RESET_TYPES:
        raise ValueError(f"Invalid robot initial state distribution. Input: {robot_reset} not in [`default`, `random`].")
    if object_reset not in capi.RESET_TYPES:
        raise ValueError(f"Invalid object initial state distribution. Input: {object_reset} "
                         "not in [`default`, `random`, `none`].")
    if task_difficulty not in (-1, 1, 2, 3, 4, 5, 6):
        raise ValueError(f"Invalid difficulty index for task: {task_difficulty}.")
    cfg = capi.TfConfig()
    cfg.api_version = capi.TF_API_VERSION
    cfg.num_envs = int(num_envs)
    cfg.env_id_offset = int(env_id_offset)
    cfg.global_num_envs = int(global_num_envs) if global_num_envs else int(num_envs)
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.command_mode = capi.COMMAND_MODES[command_mode]
    cfg.normalize_action = int(bool(normalize_action))
    cfg.normalize_obs = int(bool(normalize_obs))
    cfg.apply_safety_damping = int(bool(apply_safety_damping))
    cfg.asymmetric_obs = int(bool(asymmetric_obs))
    cfg.enable_ft_sensors = int(bool(enable_ft_sensors or asymmetric_obs))   # trifinger_env.py:272-273
    cfg.task_difficulty = int(task_difficulty)
    cfg.episode_length = int(episode_length) if episode_length else 0
    cfg.control_decimation = int(control_decimation)
    cfg.robot_reset_type = capi.RESET_TYPES[robot_reset]
    cfg.dof_pos_stddev = float(dof_pos_stddev)
    cfg.dof_vel_stddev = float(dof_vel_stddev)
    cfg.object_reset_type = capi.RESET_TYPES[object_reset]
    cfg.goal_rotation_activate = int(bool(goal_rotation))
    cfg.goal_rotation_rate_magnitude = float(goal_rotation_rate)
    fill_reward_terms(cfg, reward_terms if reward_terms is not None else default_reward_terms())
    s = {"activate": True, "bonus": 5000.0, "position_tolerance": 0.01, "orientation_tolerance": 0.2}
    s.update(success or {})
    cfg.success_activate = int(bool(s["activate"]))
    cfg.success_bonus = float(s["bonus"])
    cfg.position_tolerance = float(s["position_tolerance"])
    cfg.orientation_tolerance = float(s["orientation_tolerance"])
    cfg.dt = float(dt)
    cfg.substeps = int(substeps)
    cfg.solver_iterations = int(solver_iterations)
    cfg.solver_inner = int(solver_inner)
    for i in range(3):
        cfg.gravity[i] = float(gravity[i])
    dr = {"activate": False, "cube_mass": (0.7, 1.3), "cube_size": (0.9, 1.1), "friction": (0.7, 1.3),
          "motor_torque": (0.9, 1.1), "link_mass": (0.9, 1.1), "restitution": (0.5, 1.5), "obs_noise": 0.0, "action_repeat_prob": 0.0,
          # the rest of the reference's intent list (trifinger_env.py:387-389); neutral unless asked for
          "robot_base_position": (0.0, 0.0, 0.0), "stage_position": (0.0, 0.0),
          "friction_robot": (1.0, 1.0), "friction_object": (1.0, 1.0), "friction_stage": (1.0, 1.0)}
    dr.update(domain_randomization or {})
    cfg.dr_enable = int(bool(dr["activate"]))
    if not (float(dr["obs_noise"]) >= 0.0):
        raise ValueError(f"domain_randomization.obs_noise: need a half-width >= 0, got {dr['obs_noise']}")
    cfg.dr_obs_noise = float(dr["obs_noise"])
    if not (0.0 <= float(dr["action_repeat_prob"]) <= 1.0):
        raise ValueError(f"domain_randomization.action_repeat_prob: need a probability, got {dr['action_repeat_prob']}")
    cfg.dr_action_repeat = float(dr["action_repeat_prob"])
    for name, field in (("robot_base_position", cfg.dr_base_pos), ("stage_position", cfg.dr_stage_pos)):
        half = [float(x) for x in dr[name]]
        if len(half) != len(field) or not all(0.0 <= x <= 0.05 for x in half):
            raise ValueError(f"domain_randomization.{name}: need {len(field)} half-widths in [0, 0.05] m, got {dr[name]}")
        for k, x in enumerate(half):
            field[k] = x
    for name, field in (("cube_mass", cfg.dr_cube_mass), ("cube_size", cfg.dr_cube_size),
                        ("friction", cfg.dr_friction), ("motor_torque", cfg.dr_motor),
                        ("link_mass", cfg.dr_link_mass), ("restitution", cfg.dr_restitution),
                        ("friction_robot", cfg.dr_friction_robot), ("friction_object", cfg.dr_friction_object),
                        ("friction_stage", cfg.dr_friction_stage)):
        lo, hi = dr[name]
        if not (0.0 < float(lo) <= float(hi)):
            raise ValueError(f"domain_randomization.{name}: need 0 < lo <= hi, got {(lo, hi)}")
        field[0], field[1] = float(lo), float(hi)
    cfg.model = model if model is not None else lib.default_model()
    return cfg


_STATUS_TO_EXC = {
    capi.TF_ERR_COMMAND_MODE: (ValueError, "Invalid command mode."),
    capi.TF_ERR_ROBOT_RESET: (ValueError, "Invalid robot initial state distribution."),
    capi.TF_ERR_OBJECT_RESET: (ValueError, "Invalid object initial state distribution."),
    capi.TF_ERR_DIFFICULTY: (ValueError, "Invalid difficulty index for task."),
    capi.TF_ERR_INVALID_ARG: (ValueError, "invalid argument"),
    capi.TF_ERR_NOT_BOUND: (RuntimeError, "buffers not bound"),
    capi.TF_ERR_DEVICE: (RuntimeError, "device error"),
    capi.TF_ERR_UNSUPPORTED: (NotImplementedError, "configuration not built"),
}


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


def check(lib, status, what):
    if status == capi.TF_OK:
        return
    exc, msg = _STATUS_TO_EXC.get(status, (RuntimeError, "error"))
    raise exc(f"{what}: {msg} (status {status}; {lib.last_error()})")


class TrifingerEngine:
    """Buffers + handle for `num_envs` environments on one device."""

    def __init__(self, cfg, device="cuda:0", lib=None):
        self.device = torch.device(device)
        if lib is None:
            if self.device.type != "cuda":
                raise RuntimeError(
                    "the TriFinger step runs only as HIP kernels on an MI355X: device must be 'cuda:N' "
                    f"(got '{device}'); there is no CPU path in this package")
            lib = capi.load_hip_library()
        self.lib = lib
        self.cfg = cfg
        n = cfg.num_envs
        self.num_envs = n
        self.action_dim = lib.tf_action_dim(cfg.command_mode)
        check(lib, min(self.action_dim, 0), "tf_action_dim")
        self.obs_dim = 32 + self.action_dim
        self.states_dim = self.obs_dim + 72 if cfg.asymmetric_obs else 0
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.state = torch.zeros((capi.TF_STATE_ROWS, n), **f32)
        self.state[capi.S_CUBE_Q + 3] = 1.0     # identity quaternions (xyzw)
        self.state[capi.S_GOAL_Q + 3] = 1.0
        self.state[capi.S_PREV_OBJ_Q + 3] = 1.0
        self.state[capi.S_DR:capi.S_DR + capi.TF_NUM_DR] = 1.0    # domain-randomisation scale factors ...
        self.state[capi.S_DR + capi.DR_BASE_POS:capi.S_DR + capi.DR_FRICTION_ROBOT] = 0.0    # ... and offsets
        self.action_buf = torch.zeros((n, self.action_dim), **f32)
        self.obs = torch.zeros((n, self.obs_dim), **f32)
        self.states = torch.zeros((n, self.states_dim), **f32)
        self.reward = torch.zeros((n,), **f32)
        self.reset_buf = torch.zeros((n,), dtype=torch.bool, device=dev)
        self.goal_reset_buf = torch.zeros((n,), dtype=torch.bool, device=dev)
        self.successes = torch.zeros((n,), dtype=torch.bool, device=dev)
        self.dones = torch.zeros((n,), dtype=torch.bool, device=dev)
        self.steps = torch.zeros((n,), dtype=torch.int64, device=dev)        # torch.long like the reference's _steps_count_buf (env_base.py:572)
        self.reset_count = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.info = torch.zeros((capi.TF_NUM_INFO,), **f32)
        self.scratch = torch.zeros((int(lib.tf_scratch_floats(n)),), **f32)
        self._is_cuda = dev.type == "cuda"
        self._dev_type = dev.type
        self._dev_index = None
        if self._is_cuda:
            self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
            self.device = torch.device("cuda", self._dev_index)      # 'cuda' -> 'cuda:k': what tensors on it report
        self._handle = C.c_void_p()
        with self._on_device():          # the library allocates its parameter block on the CURRENT HIP device
            check(lib, lib.tf_create(C.byref(cfg), C.byref(self._handle)), "tf_create")
        b = capi.TfBuffers()
        b.state = self.state.data_ptr()
        b.action_buf = self.action_buf.data_ptr()
        b.obs = self.obs.data_ptr()
        b.states = self.states.data_ptr() if self.states_dim else None
        b.reward = self.reward.data_ptr()
        b.reset_buf = self.reset_buf.data_ptr()
        b.goal_reset_buf = self.goal_reset_buf.data_ptr()
        b.successes = self.successes.data_ptr()
        b.dones = self.dones.data_ptr()
        b.steps = self.steps.data_ptr()
        b.reset_count = self.reset_count.data_ptr()
        b.info = self.info.data_ptr()
        b.scratch = self.scratch.data_ptr()
        self._bufs = b
        with self._on_device():
            check(lib, lib.tf_bind(self._handle, C.byref(b)), "tf_bind")

    # -- plumbing ---------------------------------------------------------------------------
    def _on_device(self):
        """Context that makes the engine's GPU the current HIP device (kernels are launched on the current device)."""
        if self._is_cuda and torch.cuda.current_device() != self._dev_index:
            return torch.cuda.device(self._dev_index)
        return _NULL_CTX

    def _stream(self):
        if self._is_cuda:
            return C.c_void_p(torch.cuda.current_stream(self._dev_index).cuda_stream)
        return None

    def close(self):
        if self._handle:
            self.lib.tf_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def frame_count(self):
        return int(self.lib.tf_frame_count(self._handle))

    @frame_count.setter
    def frame_count(self, v):
        check(self.lib, self.lib.tf_set_frame_count(self._handle, int(v)), "tf_set_frame_count")

    KERNEL_VARIANTS = {"auto": 0, "narrow": 1, "wide": 2, "wide_helpers": 3}       # TF_KERNEL_* of include/trifinger.h

    @property
    def kernel_variant(self):
        """which instantiation of the fused step the launches use: 'narrow' (128 registers, four workgroups per CU), 'wide' (256
        registers, picked for num_envs <= 32768) or 'wide_helpers' (the same in workgroups of eight wavefronts - four helpers carry the finger-finger
        rows -, picked for num_envs <= 16384); same results bit for bit"""
        v = int(self.lib.tf_kernel_variant(self._handle))
        check(self.lib, min(v, 0), "tf_kernel_variant")
        return {1: "narrow", 2: "wide", 3: "wide_helpers"}[v]

    @property
    def kernel_occupancy(self):
        """workgroups of the fused step per CU (HIP runtime's figure for the instantiation this engine launches; 0 from the oracle)"""
        return int(self.lib.tf_kernel_occupancy(self._handle))

    @kernel_variant.setter
    def kernel_variant(self, name):
        check(self.lib, self.lib.tf_set_kernel_variant(self._handle, self.KERNEL_VARIANTS[name]), "tf_set_kernel_variant")

    # -- checkpoint / exact replay (the reference never checkpoints env state; optional here) -----------------
    _CHECKPOINT_BUFFERS = ("state", "action_buf", "obs", "states", "reward", "reset_buf", "goal_reset_buf", "successes", "dones",
                           "steps", "reset_count", "info")

    def state_dict(self):
        """Everything a bit-exact continuation of the rollout needs: copies of the caller-owned buffers of TfBuffers (the SoA state
        with the solver's warm start, flags, step and reset counters - the Philox counters of every sampler -, outputs of the last
        step) and the frame count (reward schedule, noise and action-repeat counters).  The handle holds nothing else that the step
        reads back: gravity, clipping and the configuration are settings of the caller."""
        d = {k: getattr(self, k).detach().clone() for k in self._CHECKPOINT_BUFFERS}
        d["frame_count"] = self.frame_count
        d["layout"] = dict(api_version=int(capi.TF_API_VERSION), num_envs=self.num_envs, action_dim=self.action_dim,
                           states_dim=self.states_dim, env_id_offset=int(self.cfg.env_id_offset), seed=int(self.cfg.seed))
        return d

    def load_state_dict(self, d):
        """Restore `state_dict()` in place (the buffers stay where the handle is bound to them).  Raises ValueError when the
        checkpoint was written by another state layout, env count, action / states width, shard offset or seed."""
        mine = dict(api_version=int(capi.TF_API_VERSION), num_envs=self.num_envs, action_dim=self.action_dim,
                    states_dim=self.states_dim, env_id_offset=int(self.cfg.env_id_offset), seed=int(self.cfg.seed))
        theirs = dict(d["layout"])
        if theirs != mine:
            raise ValueError(f"checkpoint of another engine: {theirs} (this engine: {mine})")
        for k in self._CHECKPOINT_BUFFERS:
            dst, src = getattr(self, k), d[k]
            if tuple(dst.shape) != tuple(src.shape) or dst.dtype != src.dtype:
                raise ValueError(f"checkpoint buffer '{k}': {tuple(src.shape)} {src.dtype}, expected {tuple(dst.shape)} {dst.dtype}")
        for k in self._CHECKPOINT_BUFFERS:
            getattr(self, k).copy_(d[k])
        self.frame_count = int(d["frame_count"])

    def set_clipping(self, clip_obs, clip_actions):
        """Fuse the wrapper's clamps into the step (<= 0 switches a clamp off)."""
        check(self.lib, self.lib.tf_set_clipping(self._handle, float(clip_obs), float(clip_actions)), "tf_set_clipping")

    def set_gravity(self, g):
        arr = (C.c_float * 3)(*[float(x) for x in g])
        check(self.lib, self.lib.tf_set_gravity(self._handle, arr), "tf_set_gravity")

    # -- hot path ---------------------------------------------------------------------------
    def step_random(self):
        """One fused control step with the action source fused in: every env draws 2 U[0,1) - 1 per action dimension inside the
        launch (what scripts/trifinger_random_action.py feeds the env); `action_buf` holds what was drawn."""
        if self._is_cuda and torch.cuda.current_device() != self._dev_index:
            with torch.cuda.device(self._dev_index):
                rc = self.lib.tf_step_random(self._handle, self._stream())
        else:
            rc = self.lib.tf_step_random(self._handle, self._stream())
        if rc:
            check(self.lib, rc, "tf_step_random")

    def step(self, action):
        """One fused control step.  `action`: contiguous float32 [N, A] tensor on the engine's device."""
        ad = action.device                    # a host pointer handed to the kernel would fault the GPU
        if ad.type != self._dev_type or (self._is_cuda and ad.index != self._dev_index):
            raise ValueError(f"action tensor lives on {action.device}, the engine on {self.device}")
        if self._is_cuda and torch.cuda.current_device() != self._dev_index:
            with torch.cuda.device(self._dev_index):
                rc = self.lib.tf_step(self._handle, C.c_void_p(action.data_ptr()), self._stream())
        else:
            rc = self.lib.tf_step(self._handle, C.c_void_p(action.data_ptr()), self._stream())
        if rc:
            check(self.lib, rc, "tf_step")

    def reset(self):
        with self._on_device():
            check(self.lib, self.lib.tf_reset(self._handle, self._stream()), "tf_reset")

    def enable_kernel_timing(self, max_windows, window=1):
        """One event pair per `window` consecutive launches of the fused step kernel, for at most `max_windows` windows."""
        check(self.lib, self.lib.tf_enable_kernel_timing(self._handle, int(max_windows)), "tf_enable_kernel_timing")
        check(self.lib, self.lib.tf_set_kernel_timing_window(self._handle, int(window)), "tf_set_kernel_timing_window")

    def kernel_time_ms(self):
        """(summed duration of the timed fused-step kernels in ms, number of launches); synchronises."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        check(self.lib, self.lib.tf_kernel_time_ms(self._handle, C.byref(ms), C.byref(n)), "tf_kernel_time_ms")
        return ms.value, n.value

    # -- split path (tests) -----------------------------------------------------------------
    # The fused step hands the fingertip wrench of a step to its observation phase without writing the TF_S_FT state rows; these entries use
    # the rows.  Do not mix the two forms within an episode (a post_step() after fused steps reads a stale fingertip wrench): INTEGRATION.md.
    def apply_resets(self):
        check(self.lib, self.lib.tf_apply_resets(self._handle, self._stream()), "tf_apply_resets")

    def pre_step(self):
        check(self.lib, self.lib.tf_pre_step(self._handle, self._stream()), "tf_pre_step")

    def simulate(self):
        check(self.lib, self.lib.tf_simulate(self._handle, self._stream()), "tf_simulate")

    def post_step(self):
        check(self.lib, self.lib.tf_post_step(self._handle, self._stream()), "tf_post_step")

    def finish_step(self):
        check(self.lib, self.lib.tf_finish_step(self._handle, self._stream()), "tf_finish_step")

    # -- named views of the SoA state ---------------------------------------------------------
    def view(self, row, count):
        return self.state[row:row + count]

    @property
    def q(self):
        return self.state[capi.S_Q:capi.S_Q + 9]

    @property
    def qd(self):
        return self.state[capi.S_QD:capi.S_QD + 9]

    @property
    def cube(self):
        """[13, N]: position, quaternion (xyzw), linear velocity, angular velocity."""
        return self.state[capi.S_CUBE_P:capi.S_CUBE_P + 13]

    @property
    def goal(self):
        return self.state[capi.S_GOAL_P:capi.S_GOAL_P + 7]

    @property
    def tip_pos(self):
        return self.state[capi.S_TIP_P:capi.S_TIP_P + 9]

    @property
    def tau(self):
        return self.state[capi.S_TAU:capi.S_TAU + 9]
