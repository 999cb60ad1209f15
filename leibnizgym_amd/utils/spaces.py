"""`Box` space: gym's when it is installed, otherwise a minimal stand-in with the attributes RL-Games reads
(the reference builds `gym.spaces.Box(np.full(n, -c), np.full(n, c))`, wrappers/vec_task.py:51-56)."""
import numpy as np

try:  # pragma: no cover - depends on the host image
    from gym.spaces import Box  # type: ignore
except Exception:  # gym / gymnasium absent
    try:
        from gymnasium.spaces import Box  # type: ignore
    except Exception:
        class Box:
            def __init__(self, low, high, dtype=np.float32):
                self.low = np.asarray(low, dtype=dtype)
                self.high = np.asarray(high, dtype=dtype)
                self.shape = self.low.shape
                self.dtype = np.dtype(dtype)

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

            def sample(self):
                return np.random.uniform(self.low, self.high).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))
