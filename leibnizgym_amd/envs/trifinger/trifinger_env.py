"""TriFinger manipulation task on the native HIP engine.

Mirror of the reference `TrifingerEnv` (leibnizgym/envs/trifinger/trifinger_env.py): same constructor
`TrifingerEnv(config, device, verbose, visualize)`, same default configuration dictionary, same
observation / state / action specs, same `ValueError`s.  The per-step work of the reference's hooks
(`_reset_impl` :373, `_goal_reset_impl` :425, `_pre_step` :442, `_post_step` :500, `_fill_observations_and_states`
:959, `__check_termination` :1053, samplers :1101-1265) happens inside the fused kernel
(leibnizgym_amd/csrc/trifinger_hip.hip); this class only translates the configuration and exposes buffers.
"""
import copy
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

from ... import _capi as capi
from ...engine import TrifingerEngine, make_config
from ...utils.helpers import merged, print_info
from ..env_base import IsaacEnvBase
from .utils import CuboidalObject, TrifingerDimensions

# default configuration: same keys and values as the reference (trifinger_env.py:28-115)
TRIFINGER_DEFAULT_CONFIG_DICT = {
    "episode_length": 750,
    "task_difficulty": 1,
    "enable_ft_sensors": False,
    "command_mode": "position",
    "apply_safety_damping": True,
    "asymmetric_obs": False,
    "normalize_obs": True,
    "normalize_action": True,
    "reset_distribution": {
        "robot_initial_state": {"type": "default", "dof_pos_stddev": 0.4, "dof_vel_stddev": 0.2},
        "object_initial_state": {"type": "random"},
    },
    "goal_movement": {"rotation": {"activate": False, "rate_magnitude": 0.5}},
    "reward_terms": {
        "finger_reach_object_rate": {"activate": True, "weight": -750, "norm_p": 2},
        "finger_move_penalty": {"activate": True, "weight": -0.1},
        "object_dist": {"activate": True, "weight": 2000},
        "object_rot": {"activate": True, "weight": 300},
        "object_rot_delta": {"activate": True, "weight": -250},
        "object_move": {"activate": True, "weight": -750},
    },
    "termination_conditions": {
        "success": {"activate": True, "bonus": 5000.0, "position_tolerance": 0.01, "orientation_tolerance": 0.2},
    },
    # Build-defined (the reference ships no domain randomisation: leibnizgym/dr/__init__.py is empty, the hook is
    # commented out at utils/config_utils.py:143-151).  Per-env scale factors ~ U[lo, hi] drawn at every reset.
    "domain_randomization": {
        "activate": False,
        "cube_mass": [0.7, 1.3], "cube_size": [0.9, 1.1], "friction": [0.7, 1.3], "motor_torque": [0.9, 1.1],
        "link_mass": [0.9, 1.1], "restitution": [0.5, 1.5], "obs_noise": 0.0, "action_repeat_prob": 0.0,
    },
}


class TrifingerEnv(IsaacEnvBase):
    """Three 3-DoF fingers manipulating a 65 mm cube (Real Robot Challenge tasks, difficulty 1-4)."""

    _object_dims = CuboidalObject(0.065)
    _dims = TrifingerDimensions
    _max_torque_Nm = 0.36            # reference trifinger_env.py:149
    _max_velocity_radps = 10         # :151
    _state_history_len = 2           # :228

    def __init__(self, config: dict = None, device: str = 'cpu', verbose: bool = True, visualize: bool = False, *,
                 lib=None, env_id_offset: int = 0, global_num_instances: int = None):
        """
        Args:
            config: configuration dictionary (merged over TRIFINGER_DEFAULT_CONFIG_DICT and the base defaults).
            device: torch device of every buffer; must be a 'cuda:N' device (MI355X).
            verbose / visualize: as in the reference (visualize is accepted; the env is headless).
            lib: TEST HOOK - an already loaded C-ABI library to run on instead of the HIP product library.
            env_id_offset / global_num_instances: position of this shard when envs are split over GPUs
                (the in-kernel RNG is keyed by global env id, reward schedules by global env-steps).
        """
        trifinger_config = merged(TRIFINGER_DEFAULT_CONFIG_DICT, config)
        if trifinger_config["asymmetric_obs"]:                       # trifinger_env.py:272-273
            trifinger_config["enable_ft_sensors"] = True
        if trifinger_config["command_mode"] not in capi.COMMAND_MODES:
            raise ValueError(f"Invalid command mode. Input: {trifinger_config['command_mode']} "
                             f"not in ['torque', 'position'].")
        action_dim = self._dims.JointTorqueDim.value
        if trifinger_config["command_mode"] == "position_impedance":
            action_dim *= 2
        obs_spec = {
            "robot_q": self._dims.GeneralizedCoordinatesDim.value,
            "robot_u": self._dims.GeneralizedVelocityDim.value,
            "object_q": self._dims.ObjectPoseDim.value,
            "object_q_des": self._dims.ObjectPoseDim.value,
            "command": action_dim,
        }
        if trifinger_config["asymmetric_obs"]:
            state_spec = dict(obs_spec)
            state_spec.update({
                "object_u": self._dims.ObjectVelocityDim.value,
                "fingertip_state": self._dims.NumFingers.value * self._dims.StateDim.value,
                "robot_a": self._dims.GeneralizedVelocityDim.value,
                "fingertip_wrench": self._dims.NumFingers.value * self._dims.WrenchDim.value,
            })
        else:
            state_spec = {}
        action_spec = {"command": action_dim}
        self._lib = lib
        self._env_id_offset = int(env_id_offset)
        self._global_n = global_num_instances
        self._fingertips_handles = OrderedDict.fromkeys(
            ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"], None)
        super().__init__(obs_spec, action_spec, state_spec, trifinger_config,
                         device=device, verbose=verbose, visualize=visualize)
        self._configure_mdp_spaces()
        self._build_info_items()
        if self.verbose:
            print_info("Reward terms: ")
            for name, conf in self.config["reward_terms"].items():
                print(f"\t {name}: {conf}")

    # ------------------------------------------------------------------------------------------
    def _global_num_instances(self) -> int:
        return int(self._global_n) if self._global_n else self.num_instances

    def _create_engine(self):
        c = self.config
        rd = c["reset_distribution"]
        native = c.get("native", {})
        lib = self._lib
        if lib is None:
            if not str(self.device).startswith("cuda"):
                raise RuntimeError(
                    "TrifingerEnv runs on an MI355X only: pass device='cuda:N'. There is no CPU pipeline "
                    "(the reference's sim_device='cpu' configuration has no counterpart in this package).")
            lib = capi.load_hip_library()
        if c["episode_length"] is not None and int(c["episode_length"]) < 0:
            raise ValueError("episode_length must be None or >= 0")
        # Solver: `sim.physx.solver_type` 1 asks PhysX for TGS (env_base.py:62-63), which advances the bodies after every
        # position iteration; the native step solves PGS sweeps at fixed positions by default (the north star's choice) and
        # offers the temporal form as "native.solver": "tgs" - num_position_iterations sub-steps of one sweep each with the
        # contacts regenerated every time (about 2.5x the cost of the default 2 sub-steps x 8 sweeps).
        iterations = int(c["sim"]["physx"]["num_position_iterations"])
        solver = str(native.get("solver", "pgs")).lower()
        if solver == "tgs":
            substeps, iterations = int(native.get("substeps", iterations)), 1
        elif solver == "pgs":
            substeps = int(native.get("substeps", 2))
        else:
            raise ValueError(f"native.solver: 'pgs' or 'tgs', got {solver!r}")
        model = (lib.box_model(native["object_size"], native.get("object_density", 500.0))
                 if native.get("object_size") is not None else None)
        # "native.ff_middle_pairs" (default True since API 8): finger-finger contacts between the middle link of a finger and the distal link of
        # another, beyond the three distal pairs (include/trifinger.h: TfModel.ff_middle_pairs; the reference keeps all robot links in one
        # self-colliding group, :811-812).  False: the distal pairs only - the faster step of the earlier API.
        if not native.get("ff_middle_pairs", True):
            model = model if model is not None else lib.default_model()
            model.ff_middle_pairs = 0
        cfg = make_config(
            lib, int(c["num_instances"]), seed=int(c["seed"]), env_id_offset=self._env_id_offset,
            global_num_envs=self._global_num_instances(), command_mode=c["command_mode"],
            normalize_action=c["normalize_action"], normalize_obs=c["normalize_obs"],
            apply_safety_damping=c["apply_safety_damping"], asymmetric_obs=c["asymmetric_obs"],
            enable_ft_sensors=c["enable_ft_sensors"], task_difficulty=int(c["task_difficulty"]),
            episode_length=c["episode_length"], control_decimation=int(c["control_decimation"]),
            robot_reset=rd["robot_initial_state"]["type"],
            dof_pos_stddev=rd["robot_initial_state"].get("dof_pos_stddev", 0.0),
            dof_vel_stddev=rd["robot_initial_state"].get("dof_vel_stddev", 0.0),
            object_reset=rd["object_initial_state"]["type"],
            goal_rotation=c["goal_movement"]["rotation"]["activate"],
            goal_rotation_rate=c["goal_movement"]["rotation"]["rate_magnitude"],
            reward_terms=c["reward_terms"], success=c["termination_conditions"]["success"],
            dt=float(c["sim"]["dt"]),
            # `sim.substeps` is declared but never applied by the reference (env_base.py:509-527): PhysX runs
            # with gymapi's default of 2.  "native.substeps" overrides it explicitly.
            substeps=substeps,
            solver_iterations=iterations,
            # "native.solver_inner" (default 1): passes over the block of all rows that touch the cube per sweep (include/trifinger.h: TfConfig.solver_inner)
            solver_inner=int(native.get("solver_inner", 1)),
            gravity=c["sim"]["gravity"], domain_randomization=c.get("domain_randomization"),
            # "native.object_size" (x, y, z in metres) / "native.object_density": a general box instead of the 65 mm cube,
            # e.g. [0.02, 0.08, 0.02] / 500 for objects/urdf/cube_multicolor_rrc_phase3.urdf of the reference's assets
            model=model)
        return TrifingerEngine(cfg, device=self.device, lib=lib)

    def _configure_mdp_spaces(self):
        """Scale tables exposed for inspection (the kernel holds its own copy): reference :630-748."""
        dev = self._engine.device
        t = lambda v: torch.tensor(v, dtype=torch.float, device=dev)  # noqa: E731
        q_lo, q_hi = t([-0.33, 0.0, -2.7] * 3), t([1.0, 1.57, 0.0] * 3)
        tau = t([self._max_torque_Nm] * 9)
        mode = self.config["command_mode"]
        if mode == "position":
            self._action_scale.low, self._action_scale.high = q_lo, q_hi
        elif mode == "torque":
            self._action_scale.low, self._action_scale.high = -tau, tau
        else:
            self._action_scale.low = torch.cat([q_lo, t([1.0] * 9)])
            self._action_scale.high = torch.cat([q_hi, t([50.0] * 9)])
        a_dim = self.get_action_dim()
        if self.config["normalize_action"]:
            oa = SimpleNamespace(low=t([-1.0] * a_dim), high=t([1.0] * a_dim))
        else:
            oa = self._action_scale
        pos_lo, pos_hi = t([-0.3, -0.3, 0.0]), t([0.3, 0.3, 0.3])
        ori = t([1.0] * 4)
        vel = t([float(self._max_velocity_radps)] * 9)
        self._observations_scale.low = torch.cat([q_lo, -vel, pos_lo, -ori, pos_lo, -ori, oa.low])
        self._observations_scale.high = torch.cat([q_hi, vel, pos_hi, ori, pos_hi, ori, oa.high])
        if self.config["asymmetric_obs"]:
            tip_lo = torch.cat([t([-0.4, -0.4, 0.0]), -ori, t([-0.2] * 6)])
            tip_hi = torch.cat([t([0.4, 0.4, 0.5]), ori, t([0.2] * 6)])
            self._states_scale.low = torch.cat([self._observations_scale.low, t([-0.5] * 6), tip_lo.repeat(3),
                                                -tau, t([-1.0] * 18)])
            self._states_scale.high = torch.cat([self._observations_scale.high, t([0.5] * 6), tip_hi.repeat(3),
                                                 tau, t([1.0] * 18)])
        else:
            self._states_scale.low = torch.zeros(0, device=dev)
            self._states_scale.high = torch.zeros(0, device=dev)
        state_dim = sum(self.state_spec.values())
        obs_dim = sum(self.obs_spec.values())
        action_dim = sum(self.action_spec.values())
        for name, scale, dim, got in (("Observation", self._observations_scale, obs_dim, self.get_obs_dim()),
                                     ("States", self._states_scale, state_dim, self.get_state_dim()),
                                     ("Actions", self._action_scale, action_dim, self.get_action_dim())):
            if scale.low.shape[0] != dim or scale.high.shape[0] != dim or got != dim:
                raise AssertionError(f"{name} scaling dimensions mismatch. \tLow: {scale.low.shape[0]}, "
                                     f"\tHigh: {scale.high.shape[0]}, \tExpected: {dim}.")

    def _build_info_items(self):
        """(key, device scalar) pairs of the per-step info dict: same keys as the reference
        (trifinger_env.py:554,1068,1076,1099).  Values are views of the native info buffer: no host sync."""
        info = self._engine.info
        items = []
        for k, name in enumerate(capi.REWARD_TERM_ORDER):
            if self.config["reward_terms"].get(name, {}).get("activate", False):
                items.append((f"env/rewards/{name}", info[k]))
        items.append(("env/current_position_goal/count", info[capi.INFO_POS_COUNT]))
        items.append(("env/current_orientation_goal/count", info[capi.INFO_ORI_COUNT]))
        items.append(("env/average_consecutive_success", info[capi.INFO_SUCCESS_MEAN]))
        self._info_items = items

    # ---- read-only views of the simulation state, named as in the reference ----------------------
    @property
    def _dof_position(self) -> torch.Tensor:
        return self._engine.q.T

    @property
    def _dof_velocity(self) -> torch.Tensor:
        return self._engine.qd.T

    @property
    def _object_goal_poses_buf(self) -> torch.Tensor:
        return self._engine.goal.T

    @property
    def _object_state(self) -> torch.Tensor:
        return self._engine.cube.T

    @property
    def _successes(self) -> torch.Tensor:
        return self._engine.successes


def default_config() -> dict:
    return copy.deepcopy(TRIFINGER_DEFAULT_CONFIG_DICT)
