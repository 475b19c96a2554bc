"""Checks the inequality the exact mode rests on, for EVERY evaluation of every alignment of a workload, with the
diagnostic build -DNDTPSO_VERIFY_MARGIN (ndtpso_kernels.hpp: verify_item):

    err = |fp32 score - fp64 score|  <=  B (the derived rounding-error bound of the fp32 form for that pose)
    B  <  tau / 2 = kArbRel * |gbest cost| / 2

and counts the points the fp32 loop's folded binning would file under another table entry than the reference's.

    python -m ndtpso_slam_amd.build verify                        # build container
    python scripts/verify_margin.py [--workload config3|config5|random] [--pairs N] [--seed S]   # GPU box; prints JSON
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config3", choices=["config3", "config4", "config5", "random", "converged", "short"])
    ap.add_argument("--pairs", type=int, default=0)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--repeat", type=int, default=1,
                    help="launch every run this many times with other scan pairs and PSO seeds (statistics of the binning: 1e12 point "
                         "evaluations are 364 launches of config 3); the counters accumulate over a run's launches")
    args = ap.parse_args()
    from ndtpso_slam_amd import build as b
    os.environ["NDTPSO_LIB"] = b.build_variant("verify")
    import numpy as np
    from ndtpso_slam_amd import capi, synth
    L = capi.load()
    assert hasattr(L, "ndtpso_profile_verify_margin"), "not a -DNDTPSO_VERIFY_MARGIN build"
    L.ndtpso_profile_verify_margin.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    ctx = capi.Context(0)
    rng = np.random.default_rng(args.seed)
    runs = []
    if args.workload == "config3":
        runs.append(dict(pairs=args.pairs or 512, beams=1081, cs=0.5, P=70, I=70, dev=(0.1, 0.1, 3.1415e-3), seed=2024))
    elif args.workload == "config4":   # the 4096 pairs of BASELINE config 4, as one device would see them shard by shard
        for k in range(8):
            runs.append(dict(pairs=512, beams=1081, cs=0.5, P=70, I=70, dev=(0.1, 0.1, 3.1415e-3), seed=2024, first=512 * k, total=4096))
    elif args.workload == "config5":
        runs.append(dict(pairs=args.pairs or 2, beams=2048, cs=0.25, P=2048, I=200, dev=(0.1, 0.1, 3.1415e-3), seed=21))
    elif args.workload == "short":   # short scans: the kernels that score two items per wave (eval_pair_half), swarms wide and converged
        for k, (nb, cs) in enumerate(((361, 0.5), (181, 0.5), (541, 1.0), (361, 0.3), (500, 0.5), (97, 0.5))):
            runs.append(dict(pairs=args.pairs or 300, beams=nb, cs=cs, P=(70, 30, 70, 24, 64, 17)[k], I=(40, 50, 30, 30, 20, 40)[k],
                             dev=tuple(10.0 ** -(k % 3) * np.array([0.1, 0.1, 3e-3])), seed=500 + k))
    elif args.workload == "converged":   # tight deviations: the swarm converges, near-ties everywhere
        for k in range(4):
            runs.append(dict(pairs=args.pairs or 160, beams=1081, cs=0.5, P=30, I=50, dev=tuple(10.0 ** -(k + 2) * np.array([1, 1, 0.03])), seed=90 + k))
    else:
        for k in range(8):
            runs.append(dict(pairs=args.pairs or 140, beams=int(rng.choice([361, 721, 1081, 1441, 2048])),
                             cs=float(rng.choice([0.25, 0.3, 0.5, 0.75, 1.0])), P=int(rng.integers(8, 90)), I=int(rng.integers(5, 60)),
                             dev=tuple(rng.uniform(0.02, 0.3, 2)) + (float(rng.uniform(1e-3, 0.03)),), seed=int(rng.integers(1, 1 << 30))))
    out = {"workload": args.workload, "kArbRel": 5e-6, "runs": []}
    ROW = 24
    worst = np.zeros(ROW)
    for r in runs:
        B = r["pairs"]
        assert L.ndtpso_profile_verify_margin(None, 0, 1) == 0
        handed_over = 0
        for rep in range(args.repeat):
            p = synth.make_pairs(r["pairs"], n_beams=r["beams"], seed=r["seed"] + 7919 * rep, first_pair=r.get("first", 0), total_pairs=r.get("total"))
            geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
            pose, cost, st = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, r["cs"]), (0, 0, 0), r["dev"],
                                             capi.PSOConfig.make(r["I"], r["P"]), seeds=p.seeds + 104729 * rep, mode=capi.SCORE_EXACT)
            handed_over += int(((st["status"] & 0xffff) != 0).sum())
        v = np.zeros((B, ROW))
        assert L.ndtpso_profile_verify_margin(v.ctypes.data_as(C.c_void_p), B, 0) == 0
        row = dict(r, dev=list(map(float, r["dev"])), evaluations_checked=float(v[:, 4].sum()), points_checked=float(v[:, 7].sum()),
                   max_err=float(v[:, 0].max()), max_err_over_bound=float(v[:, 1].max()), max_bound_over_half_tau=float(v[:, 2].max()),
                   max_err_over_half_tau=float(v[:, 3].max()), points_binned_differently=float(v[:, 5].sum()), max_bound=float(v[:, 6].max()),
                   alignments_flagged=handed_over, arbitrated_mean=float(st["arbitrated"].mean()), launches=args.repeat,
                   points_within_1e11_of_a_cell_edge=float(v[:, 15].sum()), points_within_1e9_of_a_cell_edge=float(v[:, 16].sum()),
                   points_whose_folded_cell_differs=float(v[:, 17].sum()), max_abs_folded_minus_reference_cells=float(v[:, 18].max()),
                   mean_abs_folded_minus_reference_cells=float(v[:, 19].sum() / max(v[:, 20].sum(), 1.0)),
                   mean_abs_cost=float(np.abs(cost).mean()),
                   evaluations_with_err_above_bound=float(v[:, 8].sum()), iteration_evaluations_with_err_above_half_tau=float(v[:, 9].sum()),
                   max_err_over_half_tau_iterations=float(v[:, 10].max()), max_err_iterations=float(v[:, 14].max()),
                   worst_err_over_bound_case=dict(zip(("fp32", "fp64", "bound"), map(float, v[int(np.argmax(v[:, 1])), 11:14]))))
        out["runs"].append(row)
        worst = np.maximum(worst, v.max(axis=0))
    out["max_err_over_bound"] = float(worst[1])
    out["max_bound_over_half_tau"] = float(worst[2])
    out["max_err_over_half_tau"] = float(worst[3])
    out["max_err_over_half_tau_iterations"] = float(worst[10])
    out["evaluations_with_err_above_bound"] = float(sum(r["evaluations_with_err_above_bound"] for r in out["runs"]))
    out["iteration_evaluations_with_err_above_half_tau"] = float(sum(r["iteration_evaluations_with_err_above_half_tau"] for r in out["runs"]))
    out["evaluations_checked"] = float(sum(r["evaluations_checked"] for r in out["runs"]))
    out["points_checked"] = float(sum(r["points_checked"] for r in out["runs"]))
    out["points_binned_differently"] = float(sum(r["points_binned_differently"] for r in out["runs"]))
    # The folded binning (gx = fma(x, C, fma(-y, S, TX))) against the reference's floor((x c - y s + tx + w/2) / cs): two rounding
    # sequences of the same real number.  A point is filed under another cell only if an integer lies between the two values, i.e.
    # with probability |folded - reference| per coordinate for coordinates spread evenly over the cell (which the measured near-edge
    # densities confirm: 2e-11 x 2 sides x 2 coordinates per point for the 1e-11 band if so).
    pts = max(out["points_checked"], 1.0)
    out["binning"] = {
        "points_within_1e-11_of_a_cell_edge": float(sum(r["points_within_1e11_of_a_cell_edge"] for r in out["runs"])),
        "points_within_1e-9_of_a_cell_edge": float(sum(r["points_within_1e9_of_a_cell_edge"] for r in out["runs"])),
        "points_whose_folded_cell_differs_from_the_reference": float(sum(r["points_whose_folded_cell_differs"] for r in out["runs"])),
        "max_abs_folded_minus_reference_cells": float(max(r["max_abs_folded_minus_reference_cells"] for r in out["runs"])),
        "mean_abs_folded_minus_reference_cells": float(np.mean([r["mean_abs_folded_minus_reference_cells"] for r in out["runs"]])),
    }
    b = out["binning"]
    b["near_edge_density_per_point_1e-9_band"] = b["points_within_1e-9_of_a_cell_edge"] / pts
    b["expected_for_uniform_fractions_1e-9_band"] = 4e-9
    b["implied_misbin_probability_per_point_evaluation"] = 2.0 * b["mean_abs_folded_minus_reference_cells"]
    b["implied_misbins_per_alignment_of_config_3"] = b["implied_misbin_probability_per_point_evaluation"] * 4971 * 1081
    print(json.dumps(out))


if __name__ == "__main__":
    main()
