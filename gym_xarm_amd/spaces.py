"""Minimal gym.spaces stand-ins (gym / gymnasium are not installed here).  When a real `gym` or
`gymnasium` is importable its spaces are used instead, so SB3-style callers see genuine spaces.
Mirrors what the reference builds in xarm_pick_and_place.py:95-100."""
import numpy as np

try:  # pragma: no cover - not available in the build container
    from gymnasium import spaces as _real
except Exception:  # noqa
    try:
        from gym import spaces as _real
    except Exception:  # noqa
        _real = None

if _real is not None:  # pragma: no cover
    Box = _real.Box
    Dict = _real.Dict
else:
    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            if x.shape != self.shape:
                return False
            if not np.can_cast(x.dtype, self.dtype, casting="same_kind"):
                return False
            return bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)

    class Dict:
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x):
            return isinstance(x, dict) and set(x.keys()) == set(self.spaces.keys()) and all(
                self.spaces[k].contains(x[k]) for k in self.spaces)

        def __repr__(self):
            return "Dict(%s)" % ", ".join("%s: %r" % kv for kv in self.spaces.items())
