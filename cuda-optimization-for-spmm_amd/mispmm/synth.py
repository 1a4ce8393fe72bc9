"""Seeded synthetic dense B (the reference has no K parameter: its B is whatever
`dense.in` holds, src/main.cu:185; data/large_25605/dense.mtx is missing from the
checkout, so every large config uses this generator).

Counter-based, so host C++ (host/src/synth.cpp), Python tests and bench.py
produce bit-identical matrices without files:
    h      = splitmix64(seed * 2^40 + i * cols + j)
    uniform: B[i,j] = (h >> 40) / 2^23 - 1          in [-1, 1), 24-bit grid
    exact  : B[i,j] = ((h >> 55) - 256) / 256       multiples of 2^-8 in [-1, 1)
The `exact` grid makes every partial sum of a +-1-valued A (n4c6-b13) exactly
representable in fp32, so any accumulation order gives the same bits.
"""
import numpy as np

DEFAULT_SEED = 20241218

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def dense_b(rows, cols, seed=DEFAULT_SEED, mode="uniform"):
    """Row-major [rows, cols] float32."""
    with np.errstate(over="ignore"):
        idx = np.arange(rows * cols, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))
    h = splitmix64(idx)
    if mode == "uniform":
        v = (h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    elif mode == "exact":
        v = ((h >> np.uint64(55)).astype(np.int32) - 256).astype(np.float32) * np.float32(2.0 ** -8)
    else:
        raise ValueError(f"unknown synthetic mode {mode!r}")
    return v.reshape(rows, cols)


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    return r.view(np.float32)
