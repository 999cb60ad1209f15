"""Base class of the vectorised environments: configuration merge, buffer access, step/reset sequencing.

Host-side mirror of the reference's `IsaacEnvBase` (leibnizgym/envs/env_base.py): same constructor
arguments, same properties and getters, same exceptions, same (obs, reward, dones, info) contract.
What sits underneath is different: there is no simulator object - `step()` is ONE native call
(`tf_step`, include/trifinger.h) that runs the whole reference sequence of env_base.py:370-399 as a
fused HIP launch on torch's current stream, with no device->host synchronisation (the reference
synchronises at least four times per step through `torch.nonzero` / `.cpu()`).
"""
import os
import random
import sys
from types import SimpleNamespace
from typing import Dict, Tuple, Union

import numpy as np
import torch
import yaml

from ..utils.helpers import merged, print_dict, print_info, print_warn

# default configuration: same keys and values as the reference (env_base.py:30-77)
ISAACGYM_DEFAULT_CONFIG_DICT = {
    "seed": 0,
    "num_instances": 1,
    "spacing": 1.0,
    "control_decimation": 1,
    "episode_length": None,
    "aggregate_mode": True,
    "physics_engine": "physx",
    "sim": {
        "dt": 0.02,
        "substeps": 2,
        "up_axis": "z",
        "gravity": [0.0, 0.0, -9.81],
        "num_client_threads": 0,
        "use_gpu_pipeline": False,
        "physx": {
            "solver_type": 1,
            "num_position_iterations": 4,
            "num_velocity_iterations": 0,
            "num_threads": 4,
            "use_gpu": False,
            "num_subscenes": 0,
            "max_gpu_contact_pairs": 8 * 1024 * 1024,
        },
        "flex": {
            "shape_collision_margin": 0.01,
            "num_outer_iterations": 4,
            "num_inner_iterations": 10,
        },
    },
}


class IsaacEnvBase:
    """Base of the HIP-backed vectorised environments (name kept from the reference so that wrappers'
    `isinstance(task, IsaacEnvBase)` checks read the same)."""

    def __init__(self, obs_spec: Dict[str, int], action_spec: Dict[str, int], state_spec: Dict[str, int],
                 config: dict = None, device: str = 'cpu', verbose: bool = True, visualize: bool = False):
        self.obs_spec = obs_spec
        self.action_spec = action_spec
        self.state_spec = state_spec
        self.device = device
        self.verbose = verbose
        self.visualize = visualize
        self.config = merged(ISAACGYM_DEFAULT_CONFIG_DICT, config)
        if self.verbose:
            print_info("Environment configuration: ")
            print_dict(self.config, nesting=0)
            print('-' * 40)
        self.num_instances = self.config["num_instances"]
        self.control_decimation = self.config["control_decimation"]
        self.episode_length = self.config["episode_length"]
        self._observations_scale = SimpleNamespace(low=None, high=None)
        self._states_scale = SimpleNamespace(low=None, high=None)
        self._action_scale = SimpleNamespace(low=None, high=None)
        self._step_info: Dict[str, torch.Tensor] = {}
        self._validate_sim_config()
        # the native engine (buffers + handle) is created by the task
        self._engine = self._create_engine()
        self.seed(self.config["seed"])

    # ------------------------------------------------------------------------------------------
    def _validate_sim_config(self):
        # same checks and exceptions as the reference (env_base.py:504-507, 581-587)
        if self.config["physics_engine"] not in ("physx", "flex"):
            raise ValueError(f"Invalid physics engine backend: {self.config['physics_engine']}")
        if self.config["sim"]["up_axis"] not in ["z", "y"]:
            raise ValueError(f"Invalid physics up-axis: {self.config['sim']['up_axis']}")
        if self.config["sim"]["up_axis"] != "z":
            raise NotImplementedError("the native TriFinger model is z-up (every shipped config uses 'z')")

    def _create_engine(self):
        raise NotImplementedError

    # ---- configuration ------------------------------------------------------------------------
    def set_gravity(self, gravity: Tuple[float, float, float] = (0, 0, -9.81)):
        self.config["sim"]["gravity"] = [float(g) for g in gravity]
        self._engine.set_gravity(gravity)

    def get_gravity(self) -> np.ndarray:
        return np.asarray(self.config["sim"]["gravity"], dtype=np.float64)

    def get_sim_params(self) -> dict:
        return self.config["sim"]

    def set_camera_lookat(self, pos, target):
        pass  # no viewer (headless only)

    # ---- shapes -------------------------------------------------------------------------------
    def get_state_shape(self) -> torch.Size:
        return self._engine.states.size()

    def get_obs_shape(self) -> torch.Size:
        return self._engine.obs.size()

    def get_action_shape(self) -> torch.Size:
        return self._engine.action_buf.size()

    def get_num_instances(self) -> int:
        return self.num_instances

    def get_state_dim(self) -> int:
        return self.get_state_shape()[1]

    def get_obs_dim(self) -> int:
        return self.get_obs_shape()[1]

    def get_action_dim(self) -> int:
        return self.get_action_shape()[1]

    # ---- live buffers (callers get references, never copies: reference env_base.py:261-289) --------
    @property
    def states_buf(self) -> torch.Tensor:
        return self._engine.states

    @property
    def obs_buf(self) -> torch.Tensor:
        return self._engine.obs

    @property
    def action_buf(self) -> torch.Tensor:
        return self._engine.action_buf

    @property
    def reward_buf(self) -> torch.Tensor:
        return self._engine.reward

    @property
    def dones_buf(self) -> torch.Tensor:
        # the reference returns `_reset_buf` here (env_base.py:281-284)
        return self._engine.reset_buf

    # protected names used by the reference's task code and its tests
    @property
    def _reset_buf(self):
        return self._engine.reset_buf

    @property
    def _goal_reset_buf(self):
        return self._engine.goal_reset_buf

    @property
    def _steps_count_buf(self):
        return self._engine.steps

    @property
    def _obs_buf(self):
        return self._engine.obs

    @property
    def _states_buf(self):
        return self._engine.states

    @property
    def _action_buf(self):
        return self._engine.action_buf

    @property
    def _reward_buf(self):
        return self._engine.reward

    @property
    def env_steps_count(self) -> int:
        """Total number of env steps aggregated over the parallel envs: frames x instances (env_base.py:287-289)."""
        return self._engine.frame_count * self._global_num_instances()

    def _global_num_instances(self) -> int:
        return self.num_instances

    # ---- operations ---------------------------------------------------------------------------
    def dump_config(self, filename: str):
        if not filename.endswith('.yaml'):
            filename += '.yaml'
        dir_name = os.path.dirname(filename)
        if dir_name:
            os.makedirs(dir_name, exist_ok=True)
        with open(filename, 'w') as file:
            yaml.dump(self.config, file)

    @staticmethod
    def seed(seed: int = None):
        random.seed(seed)
        np.random.seed(seed if seed is None else int(seed) % (2 ** 32))
        if seed is not None:
            torch.manual_seed(seed)

    def reset(self) -> torch.Tensor:
        """Reset every env, zero action, ONE simulate, fill the observation buffers; returns a clone
        (env_base.py:322-343)."""
        self._engine.reset()
        return self._engine.obs.clone().detach()

    def step(self, action: Union[np.ndarray, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, dict]:
        """Apply `action` [N, A]; returns (obs, rewards, dones, info) - the live buffers, as the reference does."""
        if isinstance(action, np.ndarray):
            action = torch.tensor(action, dtype=torch.float, device=self.device)
        action_shape = (self.num_instances, self.get_action_dim())
        if tuple(action.size()) != action_shape:
            msg = f"Invalid shape for tensor `action`. Input: {tuple(action.size())} != {action_shape}."
            raise ValueError(msg)
        eng = self._engine
        if action.dtype != torch.float32 or action.device != eng.device or not action.is_contiguous():
            action = action.to(device=eng.device, dtype=torch.float32).contiguous()
        eng.step(action)
        self._step_info = dict(self._info_items)
        return eng.obs, eng.reward, eng.dones, self._step_info

    def fuse_clipping(self, clip_obs: float, clip_actions: float) -> bool:
        """Let the native step do the wrapper's clamps (leibnizgym/wrappers/vec_task.py:146-170): actions are limited to
        +-clip_actions as they are read, every emitted obs / states value to +-clip_obs.  Returns True when the engine
        took them over; `VecTaskPython` then passes tensors through untouched.  Side effect to know about: `obs_buf` /
        `states_buf` then hold the clipped values (the reference keeps the unclipped ones inside the task and clips a
        copy) - they differ only beyond +-clip_obs, five times the nominal range of a scaled observation."""
        self._engine.set_clipping(clip_obs, clip_actions)
        return True

    def state_dict(self) -> dict:
        """Checkpoint of the simulation (not in the reference, whose env state lives inside IsaacGym and is never saved; SURVEY.md
        section 5 lists it as optional): buffers + counters from which `load_state_dict` continues the rollout bit for bit."""
        return self._engine.state_dict()

    def load_state_dict(self, state: dict):
        self._engine.load_state_dict(state)
        self._step_info = dict(self._info_items)

    def render(self):
        if self.visualize:
            print_warn("render(): the HIP environment is headless; no viewer is available.")

    def close(self):
        if getattr(self, "_engine", None) is not None:
            self._engine.close()

    # hooks of the reference's template-method design, mapped onto the split native path (tests, debugging)
    def _reset_impl(self, instances: torch.Tensor):
        self._engine.reset_buf[instances] = True
        self._engine.apply_resets()

    def _goal_reset_impl(self, instances: torch.Tensor):
        self._engine.goal_reset_buf[instances] = True
        self._engine.apply_resets()

    def _pre_step(self):
        self._engine.pre_step()

    def _post_step(self):
        self._engine.post_step()
        self._step_info = dict(self._info_items)

    def _fill_observations_and_states(self):
        raise NotImplementedError("observations are filled by the fused step; use step()/reset()")


# keep `python -m` friendliness of the reference module
if __name__ == "__main__":  # pragma: no cover
    print(ISAACGYM_DEFAULT_CONFIG_DICT, file=sys.stderr)
