"""ORACLE tooling — a platform-independent tensor generator (numpy integer hash, no RNG state).

Golden fixtures must not depend on torch/numpy RNG streams (they differ across builds), and the
full-width weights (fc6 is 51 MB) are too large to commit.  Inputs and weights of every fixture
are therefore *functions of (seed, index)* computed with exact integer arithmetic (splitmix64) and
IEEE-exact float conversions; the fixtures store only the expected outputs.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _u24(n: int, seed: int, stream: int) -> np.ndarray:
    """n uniform integers in [0, 2^24) as float64 (exact)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        h = _splitmix64(idx ^ key)
    return (h >> np.uint64(40)).astype(np.float64)


def uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    u = _u24(n, seed, 0) / 16777216.0                       # exact in float64
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(shape, seed: int, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Irwin-Hall(4) scaled to unit variance: bell-shaped, support +-3.46 sigma, exact arithmetic."""
    n = int(np.prod(shape))
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += _u24(n, seed, k + 1) / 16777216.0
    x = (s - 2.0) * 1.7320508075688772                      # var(IH4) = 1/3
    return (mean + std * x).astype(np.float32).reshape(shape)
