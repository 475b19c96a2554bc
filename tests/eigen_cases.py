"""Covariance matrices of NDT cells drawn from the synthetic world (ndtpso_slam_amd/synth.py), for the comparison of
the eigenvalue variants the oracle carries (Eigen 3.3.7's RealSchur path / the same unscaled / the closed form).

One matrix per built cell (count > 2, ndtcell.cpp:43) of each scan: points binned with getCellIndex's arithmetic
(ndtframe.cpp:240-249), population covariance about the cell mean (ndtcell.cpp:44-55, 94).  Vectorised, so the sums
are not in the reference's order -- irrelevant here: the matrices only have to be the kind the solver meets.
"""
import numpy as np

from ndtpso_slam_amd import synth


def scan_points(ranges, amin, ainc, rmax, eps=0.1):
    r = np.asarray(ranges, dtype=np.float32)
    keep = (r > 0) & (r < np.float32(rmax)) & (r > np.float32(eps))
    th = (np.arange(r.size, dtype=np.float32) * np.float32(ainc) + np.float32(amin)).astype(np.float64)
    return np.stack([r[keep].astype(np.float64) * np.cos(th[keep]), r[keep].astype(np.float64) * np.sin(th[keep])], 1)


def cell_covariances(pts, frame_m, cs):
    W = int(np.ceil(frame_m / cs))
    half = frame_m / 2.0
    inside = (np.abs(pts[:, 0]) < half) & (np.abs(pts[:, 1]) < half)
    pts = pts[inside]
    key = (np.floor((pts[:, 0] + half) / cs) + W * np.floor((pts[:, 1] + half) / cs)).astype(np.int64)
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    sx = np.bincount(inv, pts[:, 0]) / cnt
    sy = np.bincount(inv, pts[:, 1]) / cnt
    d0, d1 = pts[:, 0] - sx[inv], pts[:, 1] - sy[inv]
    c00 = np.bincount(inv, d0 * d0) / cnt
    c01 = np.bincount(inv, d0 * d1) / cnt
    c11 = np.bincount(inv, d1 * d1) / cnt
    built = cnt > 2
    return np.stack([c00, c01, c01, c11], 1)[built]


def world_covariances(n_matrices, cell_sides=(0.5, 0.25, 0.3), frame_m=60, seed=99, chunk=64):
    """>= n_matrices covariance matrices: scans of the synthetic run (both scans of each pair), the cell sides of
    BASELINE's configurations and a non-power-of-two one."""
    out, have, first = [], 0, 0
    while have < n_matrices:
        p = synth.make_pairs(chunk, seed=seed, first_pair=first, total_pairs=4096)
        first += chunk
        for b in range(chunk):
            for r in (p.ref_ranges[b], p.new_ranges[b]):
                pts = scan_points(r, p.angle_min, p.angle_inc, p.range_max)
                for cs in cell_sides:
                    m = cell_covariances(pts, frame_m, cs)
                    out.append(m)
                    have += m.shape[0]
            if have >= n_matrices:
                break
    return np.concatenate(out)[:n_matrices]


def ulp_diff(a, b):
    """|a - b| in units of the last place of b (finite, same sign expected)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ia = a.view(np.int64)
    ib = b.view(np.int64)
    return np.abs(ia - ib)
