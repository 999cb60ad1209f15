"""Vectorised-task wrappers for RL training (counterpart of reference leibnizgym/wrappers/vec_task.py).

`VecTaskPython.step` is: clamp actions to +-clip_actions -> task.step -> clamp obs to +-clip_obs -> move to
the RL device (vec_task.py:157-170); `get_state` clamps the states the same way (:146-147)."""
from typing import Tuple

import numpy as np
import torch

from ..envs.env_base import IsaacEnvBase
from ..utils.spaces import Box


class VecTask:
    def __init__(self, task: IsaacEnvBase, rl_device: str, clip_obs: float = 5.0, clip_actions: float = 1.0):
        assert isinstance(task, IsaacEnvBase)
        self._task = task
        self._clip_obs = float(clip_obs)
        self._clip_actions = float(clip_actions)
        self._rl_device = rl_device
        self._obs_space = Box(np.full(self.num_obs, -self._clip_obs), np.full(self.num_obs, self._clip_obs))
        self._state_space = Box(np.full(self.num_states, -self._clip_obs), np.full(self.num_states, self._clip_obs))
        self._act_space = Box(np.full(self.num_actions, -self._clip_actions),
                              np.full(self.num_actions, self._clip_actions))

    def __str__(self) -> str:
        return (f"Vectorized Environment around task: {type(self._task).__name__} \n"
                f"\t Number of instances   : {self.num_envs} \n"
                f"\t Number of observations: {self.num_obs} \n"
                f"\t Number of states      : {self.num_states} \n"
                f"\t Number of actions     : {self.num_actions} \n"
                f"\t Observation clipping  : {self._clip_obs} \n"
                f"\t Actions clipping      : {self._clip_actions} \n")

    def get_number_of_agents(self) -> int:
        if hasattr(self._task, 'get_number_of_agents'):
            return self._task.get_number_of_agents()
        return 1

    @property
    def num_envs(self) -> int:
        return self._task.get_num_instances()

    @property
    def num_states(self) -> int:
        return self._task.get_state_dim()

    @property
    def num_obs(self) -> int:
        return self._task.get_obs_dim()

    @property
    def num_actions(self) -> int:
        return self._task.get_action_dim()

    @property
    def observation_space(self):
        return self._obs_space

    @property
    def state_space(self):
        return self._state_space

    @property
    def action_space(self):
        return self._act_space

    def dump_config(self, filename: str):
        self._task.dump_config(filename)

    def reset(self) -> torch.Tensor:
        raise NotImplementedError

    def step(self, actions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, dict]:
        raise NotImplementedError


class VecTaskPython(VecTask):
    def get_state(self) -> torch.Tensor:
        return torch.clamp(self._task.states_buf, -self._clip_obs, self._clip_obs).to(self._rl_device)

    def reset(self) -> torch.Tensor:
        obs = self._task.reset()
        return torch.clamp(obs, -self._clip_obs, self._clip_obs).to(self._rl_device)

    def step(self, actions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, dict]:
        if self._task.visualize:
            self._task.render()
        actions_tensor = torch.clamp(actions, -self._clip_actions, self._clip_actions)
        obs, rew, is_done, info = self._task.step(actions_tensor)
        obs = torch.clamp(obs, -self._clip_obs, self._clip_obs).to(self._rl_device)
        rew = rew.to(self._rl_device)
        is_done = is_done.to(self._rl_device)
        return obs, rew, is_done, info
