"""Regenerates tests/golden/*.npz.

The reference cannot be compiled in this image (needs Eigen3) and ships no vectors of its own, so these
fixtures are produced by the repo's CPU oracle (oracle/ndtpso_oracle.c) -- they pin the oracle against
drift and give the GPU tests fixed inputs/outputs; they are NOT reference outputs ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ndtpso_slam_amd import synth  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

FRAME, DEV = 60, (0.1, 0.1, 3.1415e-3)


def main():
    p = synth.make_pairs(4, n_beams=361, seed=99)
    out = dict(ref_ranges=p.ref_ranges, new_ranges=p.new_ranges, angle_min=p.angle_min, angle_inc=p.angle_inc,
               range_max=p.range_max, seeds=p.seeds, delta=p.delta)
    # G1 scan -> points, G2 cell tables (0.5 m and 0.25 m), G3 costs, G4 PSO poses
    ref_pts, new_pts = [], []
    for b in range(4):
        f = O.Frame((0, 0, 0), FRAME, FRAME, float(FRAME))
        f.load_laser(p.ref_ranges[b], p.angle_min, p.angle_inc, p.range_max)
        ref_pts.append(f.points())
        g = O.Frame((0, 0, 0), FRAME, FRAME, float(FRAME))
        g.load_laser(p.new_ranges[b], p.angle_min, p.angle_inc, p.range_max)
        new_pts.append(g.points())
    out["g1_ref_points_0"] = ref_pts[0]
    out["g1_new_points_0"] = new_pts[0]
    for cs, tag in ((0.5, "050"), (0.25, "025")):
        ref = O.Frame((0, 0, 0), FRAME, FRAME, cs)
        ref.load_laser(p.ref_ranges[0], p.angle_min, p.angle_inc, p.range_max)
        ref.build()
        cells = ref.cells()
        out[f"g2_{tag}_index"] = np.array([c["index"] for c in cells], dtype=np.int32)
        out[f"g2_{tag}_count"] = np.array([c["count"] for c in cells], dtype=np.int32)
        out[f"g2_{tag}_built"] = np.array([c["built"] for c in cells], dtype=np.int8)
        out[f"g2_{tag}_mean"] = np.array([c["mean"] if c["built"] else (0, 0) for c in cells])
        out[f"g2_{tag}_icov"] = np.array([c["icov"] if c["built"] else (0, 0, 0, 0) for c in cells])
    rng = np.random.default_rng(3)
    poses = p.delta[0] + rng.uniform(-1, 1, (64, 3)) * np.array([0.15, 0.15, 0.03])
    ref = O.Frame((0, 0, 0), FRAME, FRAME, 0.5)
    ref.load_laser(p.ref_ranges[0], p.angle_min, p.angle_inc, p.range_max)
    new = O.Frame((0, 0, 0), FRAME, FRAME, float(FRAME))
    new.load_laser(p.new_ranges[0], p.angle_min, p.angle_inc, p.range_max)
    out["g3_poses"] = poses
    out["g3_costs"] = np.array([ref.cost(q, new) for q in poses])
    g4 = []
    for (P, I) in ((30, 50), (70, 70)):
        for b in range(4):
            ref = O.Frame((0, 0, 0), FRAME, FRAME, 0.5)
            ref.load_laser(p.ref_ranges[b], p.angle_min, p.angle_inc, p.range_max)
            new = O.Frame((0, 0, 0), FRAME, FRAME, float(FRAME))
            new.load_laser(p.new_ranges[b], p.angle_min, p.angle_inc, p.range_max)
            pose, cost, st = ref.pso((0, 0, 0), new, DEV, O.PSOConfig.make(I, P), seed=int(p.seeds[b]))
            g4.append([P, I, b, pose[0], pose[1], pose[2], cost, st["gbest_updates"], st["pbest_updates"]])
    out["g4_pso"] = np.array(g4)
    out["glibc_rand_seed42"] = O.glibc_rand(42, 64)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"))


if __name__ == "__main__":
    main()
