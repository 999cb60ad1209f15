"""RL-facing wrapper of a task: action / observation clipping, device hand-over, gym-style spaces.

API counterpart of the reference's `VecTask` / `VecTaskPython` (leibnizgym/wrappers/vec_task.py:26-170): same
constructor, same properties (`num_envs`, `num_obs`, `num_states`, `num_actions`, `observation_space`, `state_space`,
`action_space`), same call contract.  Per control step the reference's wrapper adds three element-wise operations around
the task's step: actions are limited to +-clip_actions before the task sees them, observations (and the privileged
states returned by `get_state`) to +-clip_obs afterwards, and results are moved to the learner's device.  At 65536 envs
those clamps cost 19 us per step as separate launches (30 % on top of the 63 us step), so by default the wrapper hands
its bounds to the task (`fuse_clipping`) and the native kernel applies them as it reads the action tile and emits the
observation tiles; `fuse_clipping=False` restores the literal three-operation form.
"""
from typing import Dict, Tuple

import numpy as np
import torch

from ..envs.env_base import IsaacEnvBase
from ..utils.spaces import Box


def _symmetric_box(dim: int, bound: float) -> Box:
    edge = np.full(dim, bound)
    return Box(-edge, edge)


class VecTask:
    """Holds the task, the clipping bounds and the three spaces; `reset`/`step` are provided by subclasses."""

    def __init__(self, task: IsaacEnvBase, rl_device: str, clip_obs: float = 5.0, clip_actions: float = 1.0,
                 fuse_clipping: bool = True):
        assert isinstance(task, IsaacEnvBase), "VecTask wraps environments derived from IsaacEnvBase"
        self._task, self._rl_device = task, rl_device
        self._clip_obs, self._clip_actions = float(clip_obs), float(clip_actions)
        # the native step applies the two clamps itself when the task offers it (and both bounds are real clamps)
        self._fused = bool(fuse_clipping and self._clip_obs > 0 and self._clip_actions > 0
                           and hasattr(task, "fuse_clipping") and task.fuse_clipping(self._clip_obs, self._clip_actions))
        dims = {"obs": task.get_obs_dim(), "state": task.get_state_dim(), "act": task.get_action_dim()}
        self._dims: Dict[str, int] = dims
        self._obs_space = _symmetric_box(dims["obs"], self._clip_obs)
        self._state_space = _symmetric_box(dims["state"], self._clip_obs)
        self._act_space = _symmetric_box(dims["act"], self._clip_actions)

    # sizes ------------------------------------------------------------------------------------------
    num_envs = property(lambda self: self._task.get_num_instances())
    rl_device = property(lambda self: self._rl_device)
    num_obs = property(lambda self: self._dims["obs"])
    num_states = property(lambda self: self._dims["state"])
    num_actions = property(lambda self: self._dims["act"])
    # spaces -----------------------------------------------------------------------------------------
    observation_space = property(lambda self: self._obs_space)
    state_space = property(lambda self: self._state_space)
    action_space = property(lambda self: self._act_space)

    def get_number_of_agents(self) -> int:
        return getattr(self._task, "get_number_of_agents", lambda: 1)()

    def dump_config(self, filename: str):
        self._task.dump_config(filename)

    def __str__(self) -> str:
        rows = (("Number of instances", self.num_envs), ("Number of observations", self.num_obs),
                ("Number of states", self.num_states), ("Number of actions", self.num_actions),
                ("Observation clipping", self._clip_obs), ("Actions clipping", self._clip_actions))
        body = "".join(f"\t {k:<22}: {v} \n" for k, v in rows)
        return f"Vectorized Environment around task: {type(self._task).__name__} \n{body}"

    def reset(self) -> torch.Tensor:
        raise NotImplementedError

    def step(self, actions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, dict]:
        raise NotImplementedError


class VecTaskPython(VecTask):
    """Wrapper for tasks whose buffers are torch tensors (all of them, here)."""

    def _limit_obs(self, x: torch.Tensor) -> torch.Tensor:
        if self._fused:                 # already limited by the kernel that wrote it
            return x.to(self._rl_device)
        return x.clamp(-self._clip_obs, self._clip_obs).to(self._rl_device)

    def get_state(self) -> torch.Tensor:
        return self._limit_obs(self._task.states_buf)

    def reset(self) -> torch.Tensor:
        return self._limit_obs(self._task.reset())

    def step(self, actions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, dict]:
        task = self._task
        if task.visualize:
            task.render()
        if not self._fused:
            actions = actions.clamp(-self._clip_actions, self._clip_actions)
        obs, reward, done, info = task.step(actions)
        return self._limit_obs(obs), reward.to(self._rl_device), done.to(self._rl_device), info
