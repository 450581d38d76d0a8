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


def blob_frame(width: int, height: int, seed: int, pitch: int = 11, sigma: float = 3.0) -> np.ndarray:
    """A dense lattice of Gaussian-ish blobs (one per pitch x pitch cell, amplitude 128..255 and a -1/0/+1 px
    offset drawn from the seeded stream), 8-bit-valued float32.  Roughly every second extremum candidate of such a
    frame survives the edge/contrast filter (`sift.cpp:288-346`), 0.08 survivors per pixel against 0.003 for
    `synth_frame`: the cheapest input that pushes one image past 65535 survivors, i.e. into the `u16_t size`
    truncation of `sift.cpp:41-42` (App. B-7), without the reference's O(candidates x pixels) cost running for hours.
    Integer arithmetic only, except the 16.16 fixed-point profile table whose rounding is checked to be far from a tie."""
    gw, gh = width // pitch + 3, height // pitch + 3
    gy, gx = np.meshgrid(np.arange(gh, dtype=np.uint64), np.arange(gw, dtype=np.uint64), indexing="ij")
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xB10B) * np.uint64(0xD1B54A32D192ED03)
               + gy * np.uint64(0x2545F4914F6CDD1D) + gx) & _MASK
    r = _splitmix64(key)
    amp = (128 + ((r >> np.uint64(56)).astype(np.int64) >> 1))                      # 128..255
    jx = ((r >> np.uint64(40)) % np.uint64(3)).astype(np.int64) - 1
    jy = ((r >> np.uint64(32)) % np.uint64(3)).astype(np.int64) - 1
    reach = 2 * pitch + 2
    d2max = 2 * reach * reach
    prof = 65536.0 * np.exp(-np.arange(d2max + 1, dtype=np.float64) / (2.0 * sigma * sigma))
    frac = prof - np.floor(prof)
    assert (np.abs(frac - 0.5) > 1e-6).all(), "profile table entry too close to a rounding tie"
    table = np.floor(prof + 0.5).astype(np.int64)
    xs, ys = np.arange(width, dtype=np.int64), np.arange(height, dtype=np.int64)
    cx, cy = xs // pitch, ys // pitch
    total = np.zeros((height, width), dtype=np.int64)
    for dy in (-1, 0, 1):
        by = cy + dy + 1
        for dx in (-1, 0, 1):
            bx = cx + dx + 1
            ox = ((cx + dx) * pitch + pitch // 2)[None, :] + jx[by[:, None], bx[None, :]]
            oy = ((cy + dy) * pitch + pitch // 2)[:, None] + jy[by[:, None], bx[None, :]]
            d2 = (xs[None, :] - ox) ** 2 + (ys[:, None] - oy) ** 2
            total += amp[by[:, None], bx[None, :]] * table[np.minimum(d2, d2max)]
    return np.ascontiguousarray(np.minimum(total >> 16, 255).astype(np.uint8).astype(np.float32))
