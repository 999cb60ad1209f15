"""Base type of the reward terms (counterpart of the reference's `leibnizgym/utils/mdp.py`)."""
import torch


class RewardTerm(torch.nn.Module):
    """A named, weighted, switchable reward term.  Subclasses implement `compute(...)`, which returns the already
    weighted per-env value [N]; the env adds it to the step reward only when `activate` is set."""

    name: str = ""
    weight: float = 0.0

    def __init__(self, name: str, activate: bool, weight: float, **kwargs):
        super().__init__()
        self.name, self.activate, self.weight = name, activate, weight

    def __str__(self) -> str:
        head = f"Reward name: {self.name}, enable: {self.activate}"
        return f"{head}, weight: {self.weight}" if self.activate else head

    def compute(self, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        # the reference forwards the packed tuple and dict (`self.compute(args, kwargs)`, mdp.py:55-65), which no
        # `compute` accepts - its env calls `.compute(...)` directly; here a call simply is `compute`
        return self.compute(*args, **kwargs)
