"""NDTPSO_SCORE_EXACT: the fp32-score kernels arbitrating in fp64 every pbest / gbest comparison the fp32 cost cannot
decide (ndtpso_kernels.hpp "arbitration").  Claim under test: pose AND cost equal the fp64 score mode's bit for bit --
and therefore the oracle's pose (pso_optimization, core.cpp:50-116) -- through every entry point that runs a PSO:
fused pairs (one workgroup and clusters), the staged table (`ndtpso_align`) and the resident map (`ndtpso_map_align`).
BASELINE's full-size configurations are in tests/test_gpu_fullsize.py."""
import os

import numpy as np
import pytest

from conftest import DEVIATION, FRAME_M, oracle_frames

pytestmark = pytest.mark.gpu


def _geom(p, capi):
    return capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)


def test_exact_mode_equals_fp64_mode_on_random_configurations(ctx, oracle):
    """Random (frame, cell side incl. non power-of-two, swarm, iterations, beams, guess, deviation, dropped beams):
    exact == fp64 bit for bit; fp64 == oracle to 1e-9.  Batches of 2 (clustered small batch) and of 300 (one workgroup
    per alignment).  NDTPSO_RANDOM_CASES / NDTPSO_RANDOM_SEED ask a one-off campaign for more."""
    from ndtpso_slam_amd import capi, synth
    n_cases = int(os.environ.get("NDTPSO_RANDOM_CASES", "30"))
    rng = np.random.default_rng(int(os.environ.get("NDTPSO_RANDOM_SEED", "20250929")))
    arbitrated = 0
    for case in range(n_cases):
        n_beams = int(rng.choice([90, 181, 361, 720, 1081, 1500]))
        frame = int(rng.choice([20, 40, 60, 100, 120]))
        cs = float(rng.choice([0.2, 0.25, 0.3, 0.5, 0.75, 1.0, 1.5]))
        P = int(rng.integers(3, 91))
        I = int(rng.integers(0, 40))
        B = 2 if case % 3 else 300
        p = synth.make_pairs(2, n_beams=n_beams, seed=int(rng.integers(1, 10**6)))
        ref, new = p.ref_ranges.copy(), p.new_ranges.copy()
        ref[rng.random(ref.shape) < rng.uniform(0.0, 0.3)] = 0.0
        new[rng.random(new.shape) < 0.1] = 0.0
        reps = B // 2
        ref, new = np.tile(ref, (reps, 1)), np.tile(new, (reps, 1))
        seeds = rng.integers(1, 2**31 - 1, B).astype(np.uint32)
        guess = rng.uniform(-1, 1, (B, 3)) * np.array([0.05, 0.05, 0.01])
        # tight deviations too: a converged swarm is where costs nearly tie
        dev = np.abs(rng.normal(0, 1, (B, 3))) * np.array([0.1, 0.1, 5e-3]) * float(rng.choice([1.0, 0.1, 0.01])) + 1e-7
        geom = capi.ScanGeom(n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
        grid, cfg = capi.Grid(frame, frame, cs), capi.PSOConfig.make(I, P)
        p64, c64, s64 = ctx.align_pairs(ref, new, geom, grid, guess, dev, cfg, seeds=seeds, mode=capi.SCORE_F64)
        px, cx, sx = ctx.align_pairs(ref, new, geom, grid, guess, dev, cfg, seeds=seeds, mode=capi.SCORE_EXACT)
        assert (s64["status"] == 0).all() and (sx["status"] == 0).all(), (case, sx["status"])
        assert np.array_equal(px, p64) and np.array_equal(cx, c64), (case, n_beams, frame, cs, P, I, B, np.abs(px - p64).max())
        arbitrated += int(sx["arbitrated"].sum())
        k = min(B, 4)
        want, wcost, _ = oracle.align_pairs(ref[:k], new[:k], p.angle_min, p.angle_inc, p.range_max, 0.1, frame, frame, cs,
                                            guess[:k], dev[:k], oracle.PSOConfig.make(I, P), seeds[:k])
        assert np.abs(px[:k] - want).max() < 1e-9 and np.abs(cx[:k] - wcost).max() < 1e-8, (case, n_beams, frame, cs, P, I)
    print("exact mode, %d random configurations: %d comparisons arbitrated in fp64" % (n_cases, arbitrated))
    assert arbitrated > 0   # the test must have exercised the arbitration


def test_exact_mode_staged_table_and_resident_map(ctx, oracle, pairs8):
    """`ndtpso_align` (table image staged from HBM; a cluster of workgroups by default) and `ndtpso_map_align` (the live
    node path) in exact mode against the fp64 mode and the oracle, 30 x 50 and 70 x 70, host rand() table and device
    replay of srand(seed)."""
    from ndtpso_slam_amd import capi
    p = pairs8
    geom = _geom(p, capi)
    grid = capi.Grid(FRAME_M, FRAME_M, 0.5)
    for b, (P, I) in enumerate([(30, 50), (70, 70), (30, 50), (12, 80)]):
        cfg = capi.PSOConfig.make(I, P)
        ref, new = oracle_frames(oracle, p, b)
        new_xy = new.points()
        seed = int(p.seeds[b])
        table = oracle.glibc_rand(seed, 3 + 3 * P + 6 * P * I)
        want, wcost, _ = ref.pso((0, 0, 0), new, DEVIATION, oracle.PSOConfig.make(I, P), table=table)
        ctx.ref_from_scan(grid, p.ref_ranges[b], geom)
        for kw in (dict(rand_table=table), dict(seed=seed)):
            p64, c64, _ = ctx.align(new_xy, (0, 0, 0), DEVIATION, cfg, mode=capi.SCORE_F64, **kw)
            px, cx, sx = ctx.align(new_xy, (0, 0, 0), DEVIATION, cfg, mode=capi.SCORE_EXACT, **kw)
            assert sx["status"] == 0
            assert np.array_equal(px, p64) and cx == c64, (b, P, I, kw.keys(), px - p64)
            assert np.abs(px - want).max() < 1e-9 and abs(cx - wcost) < 1e-8
        rmap = capi.ResidentMap(ctx, grid, pool_bytes=64 << 20)
        scan = capi.ResidentScan(ctx, 4096)
        scan.load_scan(p.ref_ranges[b], geom)
        rmap.insert(scan, (0, 0, 0))
        scan.load_scan(p.new_ranges[b], geom, clip=grid)
        m64, mc64, _ = rmap.align(scan, (0, 0, 0), DEVIATION, cfg, rand_table=table, mode=capi.SCORE_F64)
        mx, mcx, msx = rmap.align(scan, (0, 0, 0), DEVIATION, cfg, rand_table=table, mode=capi.SCORE_EXACT)
        assert msx["status"] == 0
        assert np.array_equal(mx, m64) and mcx == mc64
        assert np.abs(mx - want).max() < 1e-9
        rmap.close()
        scan.close()


def test_exact_mode_survives_a_swarm_of_identical_costs(ctx, oracle, pairs8):
    """A deviation of 1e-12 makes every particle score (nearly) the same: almost every comparison is a near-tie, far more
    than one evaluation group may arbitrate -- the alignment is handed to the fp64-score kernel and still equals it."""
    from ndtpso_slam_amd import capi
    p = pairs8
    geom = _geom(p, capi)
    grid, cfg = capi.Grid(FRAME_M, FRAME_M, 0.5), capi.PSOConfig.make(20, 40)
    dev = (1e-12, 1e-12, 1e-13)
    p64, c64, s64 = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_F64)
    px, cx, sx = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
    assert (sx["status"] == 0).all()
    assert np.array_equal(px, p64) and np.array_equal(cx, c64)
    want, wcost, _ = oracle.align_pairs(p.ref_ranges, p.new_ranges, p.angle_min, p.angle_inc, p.range_max, 0.1, FRAME_M, FRAME_M,
                                        0.5, (0, 0, 0), dev, oracle.PSOConfig.make(20, 40), p.seeds)
    assert np.abs(px - want).max() < 1e-9
