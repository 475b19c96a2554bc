"""Deterministic synthetic Hokuyo-like scans of a segment world.

The reference ships no recorded data (no bags, no fixtures), so the workload the
benchmark and the parity tests run on is synthesised here: a 24 m x 18 m room
with four boxes, a 1081-beam / 270 degree / 30 m sensor with 1 cm Gaussian range
noise, float32 ranges, misses reported as 0.0 (rejected by the beam filter the
reference applies in NDTFrame::loadLaser, lib/ndtpso_slam/ndtframe.cpp:165).

A "scan pair" is (scan A taken at pose_A, scan B taken at pose_B = pose_A (+) delta),
delta ~ U(+-0.08 m, +-0.08 m, +-0.02 rad).  Aligning B onto A should return ~delta.
"""
from __future__ import annotations

import dataclasses

import numpy as np

# Hokuyo UTM-30LX-like geometry
N_BEAMS = 1081
ANGLE_MIN = np.float32(-2.356194)
ANGLE_INC = np.float32(4.712389 / 1080.0)
RANGE_MAX = np.float32(30.0)


def world_segments() -> np.ndarray:
    """[M,4] array of segments (x0,y0,x1,y1): outer room + 4 boxes."""
    segs = []

    def rect(cx, cy, w, h):
        x0, x1, y0, y1 = cx - w / 2, cx + w / 2, cy - h / 2, cy + h / 2
        segs.extend([(x0, y0, x1, y0), (x1, y0, x1, y1), (x1, y1, x0, y1), (x0, y1, x0, y0)])

    rect(0.0, 0.0, 24.0, 18.0)
    rect(-6.5, 4.5, 2.0, 1.5)
    rect(7.0, 5.0, 1.5, 2.5)
    rect(6.0, -5.5, 3.0, 1.0)
    rect(-7.5, -4.0, 1.0, 3.0)
    return np.asarray(segs, dtype=np.float64)


def beam_angles(n_beams: int = N_BEAMS, angle_min=ANGLE_MIN, angle_inc=ANGLE_INC) -> np.ndarray:
    """fp32 beam angles exactly as index_to_angle computes them (core.h:40-42)."""
    idx = np.arange(n_beams, dtype=np.float32)
    return (idx * np.float32(angle_inc) + np.float32(angle_min)).astype(np.float32)


def raycast(poses: np.ndarray, n_beams: int = N_BEAMS, angle_min=ANGLE_MIN, angle_inc=ANGLE_INC,
            range_max=RANGE_MAX, segs: np.ndarray | None = None) -> np.ndarray:
    """Noise-free ranges [S, n_beams] (float64) for sensor poses [S,3]; miss -> 0."""
    segs = world_segments() if segs is None else segs
    poses = np.atleast_2d(np.asarray(poses, dtype=np.float64))
    th = beam_angles(n_beams, angle_min, angle_inc).astype(np.float64)
    ang = poses[:, 2:3] + th[None, :]                       # [S,N]
    dx, dy = np.cos(ang), np.sin(ang)
    px = segs[:, 0][None, None, :] - poses[:, 0][:, None, None]   # [S,1,M]
    py = segs[:, 1][None, None, :] - poses[:, 1][:, None, None]
    ex = (segs[:, 2] - segs[:, 0])[None, None, :]
    ey = (segs[:, 3] - segs[:, 1])[None, None, :]
    den = dx[..., None] * ey - dy[..., None] * ex           # cross(d, e)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (px * ey - py * ex) / den
        u = (px * dy[..., None] - py * dx[..., None]) / den
    ok = (np.abs(den) > 1e-12) & (t > 1e-9) & (u >= 0.0) & (u <= 1.0)
    t = np.where(ok, t, np.inf)
    r = t.min(axis=-1)
    r = np.where(r < float(range_max), r, 0.0)
    return r


@dataclasses.dataclass
class ScanPairs:
    ref_ranges: np.ndarray   # [B,N] float32 (scan A)
    new_ranges: np.ndarray   # [B,N] float32 (scan B)
    delta: np.ndarray        # [B,3] float64 ground-truth relative pose of B in A
    pose_a: np.ndarray       # [B,3]
    angle_min: np.float32
    angle_inc: np.float32
    range_max: np.float32
    seeds: np.ndarray        # [B] uint32 srand() seeds for the PSO stream

    @property
    def n_pairs(self) -> int:
        return self.ref_ranges.shape[0]

    @property
    def n_beams(self) -> int:
        return self.ref_ranges.shape[1]


def make_pairs(n_pairs: int, n_beams: int = N_BEAMS, seed: int = 2024, noise_sigma: float = 0.01,
               first_pair: int = 0, total_pairs: int | None = None) -> ScanPairs:
    """Pairs first_pair .. first_pair+n_pairs-1 of a `total_pairs`-long recorded run.

    Pair b depends only on (seed, b, total_pairs): shards generated on different
    ranks are slices of the same run.
    """
    total = (first_pair + n_pairs) if total_pairs is None else total_pairs
    idx = np.arange(first_pair, first_pair + n_pairs)
    angle_inc = np.float32(4.712389 / float(n_beams - 1))
    # smooth closed trajectory inside the room, heading along the tangent
    s = 2.0 * np.pi * idx / float(max(total, 1))
    pose_a = np.stack([6.0 * np.cos(s), 4.0 * np.sin(s) * np.cos(0.5 * s) ** 2 + 0.3 * np.sin(3 * s),
                       s + 0.5 * np.pi + 0.2 * np.sin(5 * s)], axis=1)
    delta = np.empty((n_pairs, 3))
    ref = np.empty((n_pairs, n_beams), dtype=np.float32)
    new = np.empty((n_pairs, n_beams), dtype=np.float32)
    seeds = np.empty(n_pairs, dtype=np.uint32)
    clean_a = raycast(pose_a, n_beams, ANGLE_MIN, angle_inc)
    for k, b in enumerate(idx):
        rng = np.random.default_rng([seed, int(b)])
        delta[k] = rng.uniform(-1.0, 1.0, 3) * np.array([0.08, 0.08, 0.02])
        seeds[k] = np.uint32(rng.integers(1, 2**31 - 1))
        ca, sa = np.cos(pose_a[k, 2]), np.sin(pose_a[k, 2])
        pose_b = np.array([pose_a[k, 0] + ca * delta[k, 0] - sa * delta[k, 1],
                           pose_a[k, 1] + sa * delta[k, 0] + ca * delta[k, 1],
                           pose_a[k, 2] + delta[k, 2]])
        clean_b = raycast(pose_b[None, :], n_beams, ANGLE_MIN, angle_inc)[0]
        na = rng.normal(0.0, noise_sigma, n_beams)
        nb = rng.normal(0.0, noise_sigma, n_beams)
        ref[k] = np.where(clean_a[k] > 0, clean_a[k] + na, 0.0).astype(np.float32)
        new[k] = np.where(clean_b > 0, clean_b + nb, 0.0).astype(np.float32)
    return ScanPairs(ref, new, delta, pose_a, ANGLE_MIN, angle_inc, RANGE_MAX, seeds)
