"""Host-side reward terms with the class names, constructor keywords and `compute` signatures of the reference's
`leibnizgym/envs/trifinger/rewards.py` (T7 of SURVEY.md section 8).

Inside the native step the six terms are evaluated by the HIP kernel; these torch versions serve code written against
the reference module (reward shaping experiments, offline evaluation of logged states) and are an independent statement
of the same formulas: `tests/test_rewards_host.py` pins them to the golden vectors of the reference's functions for the
difficulty-1, difficulty-4, env-default and linear-schedule configurations at five schedule steps.

State layouts: object / fingertip state = [pos 3, quat xyzw 4, lin vel 3, ang vel 3]; goal = [pos 3, quat 4].
"""
import torch

from ...utils.mdp import RewardTerm
from ...utils.torch_utils import quat_diff_rad


def linear_schedule_interpolation(step: float, sched_start: float, sched_end: float) -> float:
    """0 before `sched_start`, 1 after `sched_end`, linear in between."""
    return max(0.0, min(1.0, (step - sched_start) / (sched_end - sched_start)))


def lgsk_kernel(x: torch.Tensor, scale: float = 50.0) -> torch.Tensor:
    """Logistic kernel 1 / (e^{s x} + 2 + e^{-s x}): 1/4 at x = 0, decays like e^{-s |x|}."""
    z = x * scale
    return 1.0 / (z.exp() + 2.0 + (-z).exp())


class _Windowed(RewardTerm):
    """Terms that are switched on inside a window [start, end] of `env_steps_count` (kwargs `thresh_sched_*`);
    start == end means "always on"."""

    def _init_window(self, kwargs):
        self.sched_start = float(kwargs.pop("thresh_sched_start", 0))
        self.sched_end = float(kwargs.pop("thresh_sched_end", 0))
        self.sched_enabled = self.sched_start != self.sched_end

    def _window(self, curr_sched_step: float) -> float:
        if not self.sched_enabled:
            return 1.0
        return 1.0 if self.sched_start <= curr_sched_step <= self.sched_end else 0.0


class ObjectDistanceReward(_Windowed):
    """Pull the object towards the goal position: w dt lgsk(|p - p_goal|)."""

    def __init__(self, name: str = "object_dist", **kwargs):
        activate, weight = kwargs.pop("activate"), kwargs.pop("weight", 2000)
        super().__init__(name, activate, weight)
        self._init_window(kwargs)

    def compute(self, dt: float, curr_sched_step: float, object_state, goal_state) -> torch.Tensor:
        dist = torch.norm(object_state[:, 0:3] - goal_state[:, 0:3], p=2, dim=-1)
        return self.weight * dt * self._window(curr_sched_step) * lgsk_kernel(dist)


class ObjectMoveReward(RewardTerm):
    """Change of the object-goal distance over the step: w (|p - g| - |p_prev - g|)."""

    def __init__(self, name: str = "object_move_reward", **kwargs):
        super().__init__(name, kwargs.pop("activate"), kwargs.pop("weight", -750))

    def compute(self, object_state, last_object_state, goal_state) -> torch.Tensor:
        now = torch.norm(object_state[:, 0:3] - goal_state[:, 0:3], dim=-1)
        before = torch.norm(last_object_state[:, 0:3] - goal_state[:, 0:3], dim=-1)
        return self.weight * (now - before)


class ObjectRotationReward(_Windowed):
    """Orientation term: w dt / (scale |theta| + scale), theta = angle between object and goal orientation.
    `epsilon` is accepted (constructor and compute) and unused, as in the reference."""

    def __init__(self, name: str = "rot_dist", **kwargs):
        activate, weight = kwargs.pop("activate"), kwargs.pop("weight", 100)
        self.epsilon, self.scale = kwargs.pop("epsilon", 0.1), kwargs.pop("scale", 1.0)
        super().__init__(name, activate, weight)
        self._init_window(kwargs)

    def compute(self, dt: float, curr_sched_step: float, object_state, goal_state, epsilon: float = 0.1) -> torch.Tensor:
        theta = quat_diff_rad(object_state[:, 3:7], goal_state[:, 3:7])
        return self.weight * (self._window(curr_sched_step) * dt / (self.scale * torch.abs(theta) + self.scale))


class ObjectRotationDeltaReward(RewardTerm):
    """Change of the orientation error over the step, blended in linearly between `linear_schedule_start/end`."""

    def __init__(self, name: str = "rot_dist_delta", **kwargs):
        activate, weight = kwargs.pop("activate"), kwargs.pop("weight", 100)
        self.sched_start = float(kwargs.pop("linear_schedule_start", 0))
        self.sched_end = float(kwargs.pop("linear_schedule_end", 0))
        self.sched_enabled = self.sched_start != self.sched_end
        super().__init__(name, activate, weight)

    def compute(self, dt: float, curr_sched_step: float, object_state, last_object_state, goal_state,
                epsilon: float = 0.1) -> torch.Tensor:
        blend = linear_schedule_interpolation(curr_sched_step, self.sched_start, self.sched_end) if self.sched_enabled else 1.0
        now = torch.abs(quat_diff_rad(object_state[:, 3:7], goal_state[:, 3:7]))
        before = torch.abs(quat_diff_rad(last_object_state[:, 3:7], goal_state[:, 3:7]))
        return self.weight * (blend * (now - before))


class FingerReachObjectRatePenalty(_Windowed):
    """Rate at which the three fingertips approach the object: w sum_f (|tip_f - p| - |tip_f,prev - p_prev|)."""

    def __init__(self, name: str = "finger_reach_object_rate", **kwargs):
        self._norm_p = kwargs.pop("norm_p", 2)
        activate, weight = kwargs.pop("activate"), kwargs.pop("weight", -250)
        super().__init__(name, activate, weight)
        self._init_window(kwargs)

    def compute(self, curr_sched_step: float, fingertip_state, last_fingertip_state, object_state,
                last_object_state) -> torch.Tensor:
        now = torch.norm(fingertip_state[:, :, 0:3] - object_state[:, None, 0:3], p=self._norm_p, dim=-1)
        before = torch.norm(last_fingertip_state[:, :, 0:3] - last_object_state[:, None, 0:3], p=self._norm_p, dim=-1)
        return self.weight * self._window(curr_sched_step) * (now - before).sum(dim=-1)


class FingertipMovementPenalty(RewardTerm):
    """Squared fingertip speed from finite differences: w sum ((tip - tip_prev) / dt)^2 over 3 fingers x 3 axes."""

    def __init__(self, name: str = "finger_move_penalty", **kwargs):
        super().__init__(name, kwargs.pop("activate"), kwargs.pop("weight", -1.0e-4))

    def compute(self, dt: float, fingertip_state, last_fingertip_state) -> torch.Tensor:
        vel = (fingertip_state[:, :, 0:3] - last_fingertip_state[:, :, 0:3]) / dt
        return self.weight * vel.pow(2).reshape(-1, 9).sum(dim=-1)


REWARD_TERMS_MAPPING = {
    "finger_reach_object_rate": FingerReachObjectRatePenalty,
    "finger_move_penalty": FingertipMovementPenalty,
    "object_dist": ObjectDistanceReward,
    "object_rot": ObjectRotationReward,
    "object_rot_delta": ObjectRotationDeltaReward,
    "object_move": ObjectMoveReward,
}
