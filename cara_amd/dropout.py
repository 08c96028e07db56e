"""CPU mirror of the weight-dropout mask of the exact mode (``cara_amd/csrc/dropout_exact.hip``).

``keep(o, i)`` of linear ``linear_id`` is a counter-based hash of ``(o * in + i, seed, linear_id)``: nothing is
stored on the device, the backward regenerates the mask, and tests / the oracle rebuild it here bit for bit.
Linear ids: ``4 * layer + {0: qkv, 1: proj, 2: fc1, 3: fc2}``; ``o`` / ``i`` index the [out, in] weight.
"""
from __future__ import annotations

import numpy as np


def keep_hash_np(idx, seed: int, linear_id: int) -> np.ndarray:
    """lowbias32 finaliser, uint32 wrap-around arithmetic (== cara_weight_dropout_hash)."""
    idx = np.asarray(idx, dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = idx * np.uint32(0x9E3779B1) ^ np.uint32((seed + linear_id * 0x85EBCA77) & 0xFFFFFFFF)
        h ^= h >> np.uint32(16)
        h = h * np.uint32(0x7FEB352D)
        h ^= h >> np.uint32(15)
        h = h * np.uint32(0x846CA68B)
        h ^= h >> np.uint32(16)
    return h.astype(np.uint32)


def keep_mask(out: int, inn: int, p: float, seed: int, linear_id: int) -> np.ndarray:
    """bool [out, in]: True where the adapter element is kept (probability 1 - p)."""
    idx = np.arange(out * inn, dtype=np.uint64).astype(np.uint32)
    thresh = np.uint32(int(float(np.float32(p)) * 16777216.0))
    return ((keep_hash_np(idx, seed, linear_id) >> np.uint32(8)) >= thresh).reshape(out, inn)
