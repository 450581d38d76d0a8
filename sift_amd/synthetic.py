"""Seeded synthetic greyscale frames for tests and bench.py.

Integer value-noise on three lattices (cell 64, 16, 4 px; weights 4:2:1), bilinear interpolation
in integer arithmetic, result quantised to 8 bit and stored as float32 — the same value range
(integer-valued floats 0..255) that the reference's `vigra::importImage` of an 8-bit file yields
(/root/reference/main.cpp:52-54).  Pure numpy integer arithmetic => bit-reproducible anywhere.
"""
from __future__ import annotations

import numpy as np

_LAYERS = ((64, 4), (16, 2), (4, 1))
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _lattice(seed: int, layer: int, gw: int, gh: int) -> np.ndarray:
    iy, ix = np.meshgrid(np.arange(gh, dtype=np.uint64), np.arange(gw, dtype=np.uint64), indexing="ij")
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
               + np.uint64(layer) * np.uint64(0xD1B54A32D192ED03)
               + iy * np.uint64(0x2545F4914F6CDD1D) + ix) & _MASK
    return (_splitmix64(key) >> np.uint64(56)).astype(np.int32)  # 0..255


def synth_frame(width: int, height: int, seed: int) -> np.ndarray:
    """Return a (height, width) float32 C-contiguous frame (x fastest), values in 0..255."""
    total = np.zeros((height, width), dtype=np.int32)
    xs = np.arange(width, dtype=np.int32)
    ys = np.arange(height, dtype=np.int32)
    for layer, (cell, weight) in enumerate(_LAYERS):
        gw, gh = width // cell + 2, height // cell + 2
        lat = _lattice(seed, layer, gw, gh)
        ix, fx = xs // cell, xs % cell
        iy, fy = ys // cell, ys % cell
        # separable form of the 4-corner bilinear sum (same integers, far fewer gathers):
        # rows[g, x] = lat[g, ix]*(cell-fx) + lat[g, ix+1]*fx ; v = rows[iy]*(cell-fy) + rows[iy+1]*fy
        rows = lat[:, ix] * (cell - fx)[None, :] + lat[:, ix + 1] * fx[None, :]
        v = rows[iy] * (cell - fy)[:, None] + rows[iy + 1] * fy[:, None]
        v //= cell * cell
        v *= weight
        total += v
    return np.ascontiguousarray((total // 7).astype(np.uint8).astype(np.float32))


def synth_batch(n: int, width: int, height: int, first_seed: int = 1) -> np.ndarray:
    """(n, height, width) float32 batch; frame i uses seed first_seed + i."""
    return np.stack([synth_frame(width, height, first_seed + i) for i in range(n)])
