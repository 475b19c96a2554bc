"""Regenerates tests/golden/oracle_golden_sequence.npz: fixture G5 of SURVEY 8c -- the node's per-scan sequence
(loadLaser -> align -> update, ndtpso_slam_node.cpp:177-244) on a small synthetic run, produced by the repo's CPU
oracle (NOT reference output: "parity unpinned", see make_golden.py).  Pins the oracle's sliding-window cells,
occupancy grid, resetCells and the deviation rule against drift, and gives the resident GPU path fixed outputs.

    python tests/golden/make_golden_sequence.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ndtpso_slam_amd import synth  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

FRAME, CS, OGCS, N_BEAMS, N_SCANS, P, I, SEED = 60, 1.0, 0.25, 181, 14, 12, 15, 31


def scans():
    rng = np.random.default_rng(17)
    s = np.linspace(0.0, 0.5, N_SCANS)
    poses = np.stack([1.0 + 1.5 * s, -2.0 + 0.6 * np.sin(2.0 * s), 0.2 + 0.3 * s], axis=1)
    amin, ainc = np.float32(-2.356194), np.float32(4.712389 / (N_BEAMS - 1))
    clean = synth.raycast(poses, n_beams=N_BEAMS, angle_min=amin, angle_inc=ainc)
    r = np.where(clean > 0, clean + rng.normal(0, 0.01, clean.shape), 0.0).astype(np.float32)
    return r, amin, ainc, np.float32(30.0)


def run(ranges, amin, ainc, rmax):
    """the sequence on the oracle; returns everything the fixture stores"""
    cfg = O.PSOConfig.make(I, P)
    n_draw = 3 + 3 * P + 6 * P * I
    stream = O.glibc_rand(SEED, n_draw * N_SCANS)
    ref = O.Frame((0, 0, 0), FRAME, FRAME, CS)
    ref.enable_occupancy_grid(OGCS)
    prev = np.zeros(3)
    poses = []
    for k in range(N_SCANS):
        cur = O.Frame((0, 0, 0), FRAME, FRAME, float(FRAME))
        cur.load_laser(ranges[k], amin, ainc, rmax)
        # NDTFrame::align (ndtframe.cpp:251-266) with the frame's own configuration: deviation rule included
        pose = prev.copy() if k == 0 else ref.align(prev, cur, cfg, table=stream[(k - 1) * n_draw:k * n_draw])
        prev = pose
        ref.update(pose, cur)
        poses.append(pose)
    ref.build()
    out = dict(poses=np.array(poses))
    cells = ref.cells()
    out["cell_index"] = np.array([c["index"] for c in cells], dtype=np.int32)
    out["cell_count"] = np.array([c["count"] for c in cells], dtype=np.int32)
    out["cell_slot"] = np.array([c["slot"] for c in cells], dtype=np.int32)
    out["cell_built"] = np.array([c["built"] for c in cells], dtype=np.int8)
    out["cell_mean"] = np.array([c["mean"] if c["built"] else (0, 0) for c in cells])
    out["cell_icov"] = np.array([c["icov"] if c["built"] else (0, 0, 0, 0) for c in cells])
    og, w, h, ext = ref.occupancy_grid()
    nz = np.nonzero(og)[0]
    out["og_shape"] = np.array([w, h], dtype=np.int32)
    out["og_extent"] = np.array(ext, dtype=np.int64)
    out["og_nonzero_index"] = nz.astype(np.int32)
    out["og_nonzero_value"] = og[nz]
    pts = ref.points_all()
    out["points_count"] = np.array(len(pts))
    out["points_head"] = pts[:64]
    out["points_sum"] = pts.sum(axis=0)
    # NDTFrame::resetCells, then two more scans straight into the frame
    ref.reset_cells()
    for k in (0, 1):
        ref.load_laser(ranges[k], amin, ainc, rmax)
        ref.build()
    cells = ref.cells()
    out["reset_cell_count"] = np.array([c["count"] for c in cells], dtype=np.int32)
    out["reset_cell_built"] = np.array([c["built"] for c in cells], dtype=np.int8)
    out["reset_cell_mean"] = np.array([c["mean"] if c["built"] else (0, 0) for c in cells])
    return out


def main():
    ranges, amin, ainc, rmax = scans()
    out = run(ranges, amin, ainc, rmax)
    out.update(ranges=ranges, angle_min=amin, angle_inc=ainc, range_max=rmax,
               params=np.array([FRAME, N_BEAMS, N_SCANS, P, I, SEED], dtype=np.int32), cell_side=np.array(CS),
               og_cell_size=np.array(OGCS))
    path = os.path.join(HERE, "oracle_golden_sequence.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
