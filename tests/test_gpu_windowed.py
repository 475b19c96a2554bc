"""K3c/K3d through the C-ABI vs the oracle: binning of transformed points, sliding-window cell statistics."""
import numpy as np
import pytest

import np_ref

from conftest import FRAME_M

pytestmark = pytest.mark.gpu


def test_points_to_cells_matches_oracle(ctx, oracle):
    from ndtpso_slam_amd import capi
    rng = np.random.default_rng(2)
    xy = rng.uniform(-12, 12, (3000, 2))
    xy[:8] = [(-10.0, 0.0), (10.0, 0.0), (0.0, -10.0), (0.0, 10.0), (0.0, 0.0), (-9.9999999, -9.9999999), (0.5, 0.5), (9.999999, 9.999999)]
    for cs in (0.5, 0.3):
        for trans in (None, (0.31, -0.22, 0.013)):
            grid = capi.Grid(20, 20, cs)
            out, idx = ctx.points_to_cells(grid, xy, trans)
            f = oracle.Frame((0, 0, 0), 20, 20, cs)
            if trans is None:
                q = xy
            else:
                c, s = np_ref.cos_sin(trans[2])   # one sincos(), like the reference built by GCC
                q = np.stack([xy[:, 0] * c - xy[:, 1] * s + trans[0], xy[:, 0] * s + xy[:, 1] * c + trans[1]], axis=1)
            assert np.abs(out - q).max() < 1e-14
            want = np.array([f.get_cell_index(x, y) for x, y in out], dtype=np.int32)
            assert np.array_equal(idx, want)


def test_windowed_build_matches_oracle(ctx, oracle, pairs8):
    """Three successive update()+build() rounds on an accumulating frame: window sums, slot advance past 50
    points and the regularised inverse follow the oracle's NDTCell::build."""
    from ndtpso_slam_amd import capi
    p = pairs8
    ref = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, 1.0)
    # host-side mirror of the bookkeeping (what host/src/ndtframe.cpp does), arithmetic on the device
    state = {}     # cell index -> dict(win=CELL_WINDOW_DTYPE scalar arrays, slots, cur id, current_count, points)
    W = 60
    for rnd in range(3):
        src = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
        src.load_laser(p.ref_ranges[rnd], p.angle_min, p.angle_inc, p.range_max)
        trans = (0.05 * rnd, -0.03 * rnd, 0.01 * rnd)
        ref.update(trans, src)
        ref.build()
        pts, idx = ctx.points_to_cells(capi.Grid(FRAME_M, FRAME_M, 1.0), src.points(), trans)
        for q, k in zip(pts, idx):
            if k < 0:
                continue
            st = state.setdefault(int(k), dict(part_sum={}, part_cov={}, part_cnt={}, gsum=np.zeros(2), gcov=np.zeros(4),
                                               gcnt=0, wid=0, cur=0, pts={}, built=0, mean=np.zeros(2), icov=np.zeros(4)))
            if st["cur"] == 0:
                st["pts"][st["wid"]] = []
            st["cur"] += 1
            st["pts"].setdefault(st["wid"], []).append(q)
            st["built"] = 0
        keys = sorted(state)
        cw = np.zeros(len(keys), dtype=capi.CELL_WINDOW_DTYPE)
        off = [0]
        flat = []
        for i, k in enumerate(keys):
            st = state[k]
            wid = st["wid"]
            cw[i]["global_sum"] = st["gsum"]
            cw[i]["global_covar_sum"] = st["gcov"]
            cw[i]["slot_sum"] = st["part_sum"].get(wid, np.zeros(2))
            cw[i]["slot_covar"] = st["part_cov"].get(wid, np.zeros(4))
            cw[i]["global_count"] = st["gcnt"]
            cw[i]["slot_count"] = st["part_cnt"].get(wid, 0)
            cw[i]["current_count"] = st["cur"]
            cw[i]["built"] = st["built"]
            flat.extend(st["pts"].get(wid, []))
            off.append(len(flat))
        ctx.cells_build_windowed(cw, off, np.array(flat).reshape(-1, 2))
        for i, k in enumerate(keys):
            st = state[k]
            wid = st["wid"]
            st["gsum"], st["gcov"], st["gcnt"] = cw[i]["global_sum"].copy(), cw[i]["global_covar_sum"].copy(), int(cw[i]["global_count"])
            st["part_sum"][wid], st["part_cov"][wid], st["part_cnt"][wid] = cw[i]["slot_sum"].copy(), cw[i]["slot_covar"].copy(), int(cw[i]["slot_count"])
            if cw[i]["global_count"] > 2:
                st["built"], st["mean"], st["icov"] = 1, cw[i]["mean"].copy(), cw[i]["icov"].copy()
            if st["cur"] > 50:
                st["wid"], st["cur"] = (wid + 1) % 100, 0
        want = {c["index"]: c for c in ref.cells()}
        assert sorted(want) == keys
        advanced = 0
        for k in keys:
            w, st = want[k], state[k]
            assert w["count"] == st["gcnt"] and w["built"] == bool(st["built"])
            advanced += st["wid"] > 0
            if w["built"]:
                assert np.array_equal(w["mean"], st["mean"])
                np.testing.assert_allclose(st["icov"], w["icov"], rtol=1e-12, atol=0)
        print("round", rnd, "cells", len(keys), "with advanced slots", advanced)
    assert advanced > 0


def test_scan_to_cells_equals_two_step_route(ctx, oracle, pairs8):
    """The fused loadLaser entry gives the same points and cells as scan_to_points + points_to_cells."""
    from ndtpso_slam_amd import capi
    p = pairs8
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    for grid, trans in ((capi.Grid(60, 60, 0.5), (0, 0, 0)), (capi.Grid(20, 20, 0.3), (0.4, -0.2, 0.05))):
        xy, idx = ctx.scan_to_cells(p.new_ranges[1], geom, grid, trans)
        xy2 = ctx.scan_to_points(p.new_ranges[1], geom, trans)
        _, idx2 = ctx.points_to_cells(grid, xy2, None)
        assert np.array_equal(xy, xy2) and np.array_equal(idx, idx2)
        f = oracle.Frame((0, 0, 0), grid.width, grid.height, grid.cell_side)
        assert np.array_equal(idx, np.array([f.get_cell_index(x, y) for x, y in xy], dtype=np.int32))


def test_occupancy_grid_values_match_oracle(ctx, oracle, pairs8):
    """f-4: the occupancy-grid rasterisation of NDTFrame::build (ndtframe.cpp:79-112) on the device."""
    from ndtpso_slam_amd import capi
    p = pairs8
    seen = 0
    for (frame, cs, ogcs) in ((60, 0.5, 0.1), (40, 1.0, 0.25), (60, 1.0, 0.25), (60, 0.5, 0.2)):
        f = oracle.Frame((0, 0, 0), frame, frame, cs)
        f.enable_occupancy_grid(ogcs)
        f.load_laser(p.ref_ranges[0], p.angle_min, p.angle_inc, p.range_max)
        f.build()
        og, ogw, ogh, mm = f.occupancy_grid()
        cells = [c for c in f.cells() if c["built"]]
        vals = ctx.occupancy_values(capi.Grid(frame, frame, cs), ogcs, [c["index"] for c in cells],
                                    [c["mean"] for c in cells], [c["icov"] for c in cells])
        k = vals.shape[1]
        W = H = int(np.ceil(frame / cs))
        worst = 0
        lo = [2 ** 32 - 1, 0, 2 ** 32 - 1, 0]
        for c, v in zip(cells, vals):
            cx, cy = c["index"] % W, c["index"] // H
            for j in range(k):
                for kk in range(k):
                    if v[j, kk] >= 0:
                        ox, oy = cx * k + j, cy * k + kk
                        worst = max(worst, abs(int(og[ox + ogh * oy]) - int(v[j, kk])))
                        lo = [min(lo[0], ox), max(lo[1], ox), min(lo[2], oy), max(lo[3], oy)]
        assert worst <= 1                      # exp rounding can move 100*p across an integer
        assert tuple(lo) == mm                 # the reference's min/max sub-cell indices
        assert (vals > 0).sum() >= (og > 0).sum() - 2
        seen += int((vals >= 0).sum())
    assert seen > 100
