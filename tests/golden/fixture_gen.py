"""Deterministic, formula-defined inputs shared by ``make_golden.py`` (which runs
the reference on them, in the build container) and by the tests (which rebuild
the same inputs on the GPU box).  numpy only; no RNG library stream is relied
on -- values come from an explicit splitmix64 integer hash, so they are the
same bits on every numpy version.
"""

from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser over uint64."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hashed_u64(shape, seed: int) -> np.ndarray:
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        base = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x1000003D1)
    return splitmix64(splitmix64(base)).reshape(shape)


def trits(shape, seed: int) -> np.ndarray:
    """Entries in {-1, 0, +1} as float32."""
    return (hashed_u64(shape, seed) % np.uint64(3)).astype(np.int64).astype(np.float32) - 1.0


def exact_mips_corpus(C: int, D: int, seed: int = 7) -> np.ndarray:
    """[C, D] float32 corpus whose inner products with ``exact_mips_queries``
    are EXACT in fp32 under any summation order and pairwise DISTINCT for
    C <= 65536 (so top-K has a unique answer and bf16 storage is lossless):
    columns 0..D-4 are trits; the last three columns hold base-64/64/16 digits
    of the row index (small integers, bf16-representable)."""
    assert D >= 8
    c = trits((C, D), seed)
    i = np.arange(C, dtype=np.int64)
    c[:, D - 3] = ((i & 63) - 32).astype(np.float32)
    c[:, D - 2] = (((i >> 6) & 63) - 32).astype(np.float32)
    c[:, D - 1] = (((i >> 12) & 15) - 8).astype(np.float32)
    return c


def exact_mips_queries(B: int, D: int, seed: int = 11) -> np.ndarray:
    """[B, D] queries matching ``exact_mips_corpus``: trits, then the weights
    2^-6, 2^-12, 2^-16 on the three index-digit columns."""
    q = trits((B, D), seed)
    q[:, D - 3] = 2.0 ** -6
    q[:, D - 2] = 2.0 ** -12
    q[:, D - 1] = 2.0 ** -16
    return q


def uniform_ids(shape, high: int, seed: int) -> np.ndarray:
    return (hashed_u64(shape, seed) % np.uint64(high)).astype(np.int64)


def gaussianish(shape, seed: int) -> np.ndarray:
    """Roughly N(0,1) float32 (sum of 4 uniforms, variance-normalised); exact
    distribution is irrelevant -- only determinism matters."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        u = (hashed_u64(shape, seed * 4 + k) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        acc += u - 0.5
    return (acc * np.sqrt(3.0)).astype(np.float32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even to bf16 precision, returned as float32."""
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + np.uint64(0x7FFF) + ((b >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return r.astype(np.uint32).view(np.float32)
