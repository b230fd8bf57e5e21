"""gym.spaces.Box stand-in (the reference builds its spaces with gym, tasks/base/rl_task.py:91-96).
gymnasium / gym are used when importable so wrappers that isinstance-check keep working."""
import numpy as np

try:  # pragma: no cover - neither package is in the build image
    from gymnasium.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001
    try:
        from gym.spaces import Box  # type: ignore
    except Exception:  # noqa: BLE001
        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                low = np.asarray(low, dtype=dtype); high = np.asarray(high, dtype=dtype)
                if shape is not None:
                    low = np.broadcast_to(low, shape).copy(); high = np.broadcast_to(high, shape).copy()
                self.low, self.high, self.shape, self.dtype = low, high, tuple(low.shape), np.dtype(dtype)
                self._rng = np.random.default_rng()

            def seed(self, seed=None):
                self._rng = np.random.default_rng(seed)

            def sample(self):
                lo = np.where(np.isfinite(self.low), self.low, -1.0); hi = np.where(np.isfinite(self.high), self.high, 1.0)
                return self._rng.uniform(lo, hi).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
