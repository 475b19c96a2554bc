"""the device-resident reference frame (ndtpso_map_*, ndtpso_points_*) against the oracle's
NDTFrame -- the node's per-scan sequence loadLaser -> align -> update (ndtpso_slam_node.cpp:177-244) with the map,
its sliding windows and the loaded scan never leaving the GPU."""
import numpy as np
import pytest

import np_ref

from conftest import FRAME_M

pytestmark = pytest.mark.gpu


def _trajectory(n_scans, seed=4, step=0.6):
    from ndtpso_slam_amd import synth
    rng = np.random.default_rng(seed)
    s = np.linspace(0.0, step, n_scans)
    poses = np.stack([2.0 + 1.2 * s, -1.0 + 0.8 * np.sin(1.5 * s), 0.3 + 0.25 * s], axis=1)
    clean = synth.raycast(poses)
    return np.where(clean > 0, clean + rng.normal(0, 0.01, clean.shape), 0.0).astype(np.float32)


def _geom():
    from ndtpso_slam_amd import capi, synth
    return capi.ScanGeom(synth.N_BEAMS, synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX, 0.1)


def _compare_cells(got, want):
    assert [c["index"] for c in got] == [c["index"] for c in want]
    for g, w in zip(got, want):
        assert g["count"] == w["count"] and g["built"] == w["built"] and g["slot"] == w["slot"], (g, w)
        if w["built"]:
            # (coincident points give a zero covariance and a NaN inverse -- in the reference too; NaN == NaN here)
            assert np.array_equal(g["mean"], w["mean"], equal_nan=True) and np.array_equal(g["icov"], w["icov"], equal_nan=True), (g, w)


def test_resident_node_sequence_matches_oracle(ctx, oracle):
    """24 scans, default 30 x 50 PSO, fp64 score: poses, every cell's window state that is observable (count, slot,
    mean, inverse covariance), every stored point and the occupancy grid equal the oracle's, bit for bit."""
    from ndtpso_slam_amd import capi, synth
    n_scans, P, I, seed, cs, ogcs = 24, 30, 50, 11, 0.5, 0.1
    ranges = _trajectory(n_scans)
    geom, grid = _geom(), capi.Grid(FRAME_M, FRAME_M, cs)
    cfg, ocfg = capi.PSOConfig.make(I, P), oracle.PSOConfig.make(I, P)
    n_draw = 3 + 3 * P + 6 * P * I
    stream = oracle.glibc_rand(seed, n_draw * n_scans)

    rmap = capi.ResidentMap(ctx, grid, og_cell_size=ogcs, pool_bytes=64 << 20)
    scan = capi.ResidentScan(ctx, synth.N_BEAMS)
    ref = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, cs)
    ref.enable_occupancy_grid(ogcs)
    prev = np.zeros(3)
    oprev = np.zeros(3)
    dev = np.array([.1, .1, 3.1415e-3])
    pose_hist = [np.zeros(3)]
    for k in range(n_scans):
        scan.load_scan(ranges[k], geom, clip=grid)
        ocur = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
        ocur.load_laser(ranges[k], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
        got_pts = scan.get()
        assert got_pts.shape == ocur.points().shape and np.abs(got_pts - ocur.points()).max() < 5e-14
        # device sincos and glibc's differ in the last ulp of a few points (tests/test_gpu_parity.py pins that);
        # from here on the oracle runs on the device's points so that everything downstream can be compared exactly
        cur = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
        for q in got_pts:
            cur.add_point(q[0], q[1])
        if k == 0:
            pose, opose = prev.copy(), oprev.copy()
        else:
            table = stream[(k - 1) * n_draw:k * n_draw]
            # NDTFrame::align's deviation rule (ndtframe.cpp:253), same numbers on both sides
            d = dev if k - 1 < 2 else np.abs(2. * (pose_hist[-1] - pose_hist[-2]))
            pose, _, st = rmap.align(scan, prev, d, cfg, rand_table=table, mode=capi.SCORE_F64)
            opose, _, _ = ref.pso(oprev, cur, d, ocfg, table=table)
            assert st["n_points"] == len(cur.points())
            assert np.array_equal(pose, opose), (k, pose, opose)
            pose_hist.append(pose.copy())
        prev, oprev = pose, opose
        rmap.insert(scan, pose)
        ref.update(opose, cur)
    # the last update has not been built yet on either side (lazy build, core.cpp:27-28)
    assert np.array_equal(rmap.points(), ref.points_all())
    rmap.build()
    ref.build()
    _compare_cells(rmap.cells(), ref.cells())
    assert max(c["slot"] for c in ref.cells()) >= 1            # the window rotated somewhere
    assert np.array_equal(rmap.points(), ref.points_all())
    og, w, h, ext = rmap.occupancy()
    want, ww, wh, mm = ref.occupancy_grid()
    assert (w, h, ext) == (ww, wh, mm)
    d = np.abs(og.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).sum() <= 0.01 * max(1, (want != 0).sum())
    info = rmap.info()
    assert info["status"] == 0 and info["n_created"] == len(ref.cells())
    assert info["n_points"] == len(ref.points_all())            # nothing rotated out of a 100-slot window yet


def test_resident_map_window_wraps_and_pool_recycles(ctx, oracle):
    """A small map hammered with the same scan: cells rotate through all 100 window slots and start overwriting
    them (NDTCell::addPoint's slot reset, ndtcell.cpp:22-27), chunks return to the pool and are handed out again."""
    from ndtpso_slam_amd import capi, synth
    rng = np.random.default_rng(5)
    cs = 2.0
    grid = capi.Grid(4, 4, cs)
    rmap = capi.ResidentMap(ctx, grid, pool_bytes=8 << 20)
    ref = oracle.Frame((0, 0, 0), 4, 4, cs)
    for it in range(260):
        n = int(rng.integers(1, 500)) if it % 11 else int(rng.integers(1025, 2300))   # some inserts span several tiles
        xy = rng.uniform(-2.3, 2.3, size=(n, 2))
        xy[::7] = np.round(xy[::7])                      # points on cell edges and frame borders
        pose = (rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(-.1, .1)) if it % 3 else None
        rmap.insert_host(xy, pose)
        if pose is None:
            for p in xy:
                ref.add_point(p[0], p[1])
        else:
            c, s = np_ref.cos_sin(pose[2])   # one sincos(), like the reference built by GCC
            for p in xy:
                ref.add_point(p[0] * c - p[1] * s + pose[0], p[0] * s + p[1] * c + pose[1])
        if it % 5:
            rmap.build()
            ref.build()
        if it % 37 == 0:
            _compare_cells(rmap.cells(), ref.cells())
    rmap.build()
    ref.build()
    _compare_cells(rmap.cells(), ref.cells())
    assert np.array_equal(rmap.points(), ref.points_all())
    info = rmap.info()
    assert info["status"] == 0
    # slots were overwritten on the second lap of the window and their chunks handed out again
    assert len(ref.points_all()) < info["n_points"] and info["pool_bump"] * 32 < info["n_points"]
    assert info["pool_free"] >= 0
    # NDTFrame::resetCells keeps the window's partial terms and the built flags (ndtcell.cpp:80-91): same on the device
    rmap.reset()
    ref.reset_cells()
    assert len(rmap.points()) == 0 and rmap.info()["pool_free"] == info["pool_bump"]     # every chunk returned
    for it in range(6):
        xy = rng.uniform(-2.3, 2.3, size=(int(rng.integers(20, 300)), 2))
        rmap.insert_host(xy)
        for p in xy:
            ref.add_point(p[0], p[1])
        rmap.build()
        ref.build()
        _compare_cells(rmap.cells(), ref.cells())
    assert np.array_equal(rmap.points(), ref.points_all())
    rmap.clear()
    assert rmap.info()["n_created"] == 0 and len(rmap.points()) == 0


def test_scans_loaded_back_to_back_reuse_their_pinned_slots(ctx, oracle):
    """ndtpso_points_load_scan stages the ranges in a ring of pinned slots the kernel reads in place; a slot is free again
    once an alignment launched after its reader has reported -- or, with no alignment in between (here), once the stream
    has drained.  Twelve scans back to back into two buffers: each holds what its last scan gives."""
    from ndtpso_slam_amd import capi, synth
    ranges = _trajectory(12)
    geom, grid = _geom(), capi.Grid(FRAME_M, FRAME_M, 0.5)
    scans = [capi.ResidentScan(ctx, synth.N_BEAMS), capi.ResidentScan(ctx, synth.N_BEAMS)]
    for k in range(12):
        scans[k % 2].load_scan(ranges[k], geom, clip=grid)
    for j, k in ((0, 10), (1, 11)):
        a = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
        a.load_laser(ranges[k], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
        assert np.array_equal(scans[j].get(), a.points())


def test_resident_scan_append_and_errors(ctx, oracle):
    from ndtpso_slam_amd import capi, synth
    ranges = _trajectory(2)
    geom, grid = _geom(), capi.Grid(FRAME_M, FRAME_M, 0.5)
    scan = capi.ResidentScan(ctx, 2 * synth.N_BEAMS)
    scan.load_scan(ranges[0], geom, clip=grid)
    scan.load_scan(ranges[1], geom, trans=(0.3, -0.2, 0.05), clip=grid, append=True)
    a = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
    a.load_laser(ranges[0], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
    a.set_trans((0.3, -0.2, 0.05))
    a.load_laser(ranges[1], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
    assert scan.get().shape == a.points().shape and np.abs(scan.get() - a.points()).max() < 5e-14
    with pytest.raises(capi.NdtpsoError) as e:
        scan.load_scan(ranges[0], geom, append=True)        # third scan does not fit
    assert e.value.code == capi.E_CAPACITY
    # a point pool that is too small grows (the reference's cells grow their vectors): every point is kept ...
    n_scan = scan.get().shape[0]
    tiny = capi.ResidentMap(ctx, grid, pool_bytes=64 * 512)
    for _ in range(3):
        tiny.insert(scan, (0, 0, 0))
    assert tiny.info()["n_points"] == 3 * n_scan and tiny.points().shape[0] == 3 * n_scan
    big = capi.ResidentMap(ctx, grid, pool_bytes=64 << 20)
    for _ in range(3):
        big.insert(scan, (0, 0, 0))
    assert np.array_equal(tiny.points(), big.points())
    tiny.build()
    big.build()
    ct, cb = tiny.cells(), big.cells()
    assert len(ct) == len(cb) and all(a["index"] == b["index"] and a["count"] == b["count"] and a["built"] == b["built"] and
                                      np.array_equal(a["mean"], b["mean"], equal_nan=True) and
                                      np.array_equal(a["icov"], b["icov"], equal_nan=True) for a, b in zip(ct, cb))
    # ... and where it cannot (NDTPSO_MAP_POOL_FIXED: what a failed allocation leaves), running out is reported, not
    # silently truncated
    import os
    os.environ["NDTPSO_MAP_POOL_FIXED"] = "1"
    try:
        fixed = capi.ResidentMap(ctx, grid, pool_bytes=64 * 512)
        fixed.insert(scan, (0, 0, 0))
        with pytest.raises(capi.NdtpsoError) as e:
            fixed.info()
        assert e.value.code == capi.E_CAPACITY
    finally:
        del os.environ["NDTPSO_MAP_POOL_FIXED"]


def test_resident_frame_of_a_million_cells_built_from_end_to_end(ctx, oracle):
    """A frame of 1.44 M cells is accepted (the reference's 300 m node frame at 0.25 m), and so is a BUILT box that spans it from
    corner to corner -- a robot that has been everywhere: the packer assembles such a box's bitmap (45 000 words, more than LDS
    holds) in the table's image in HBM, and the alignment reads the table through L2.  Poses equal the oracle's; until round 6 the
    frame was refused at creation, then (for a few hours of that round) its alignments once the box outgrew LDS.  A frame of 2^21
    cells or more is refused at creation (the drop-in keeps such a frame on the host's side: tests/test_host_library.py)."""
    from ndtpso_slam_amd import capi, synth
    grid = capi.Grid(300, 300, 0.25)
    m = capi.ResidentMap(ctx, grid)
    geom = _geom()
    scan = capi.ResidentScan(ctx, synth.N_BEAMS)
    scan.load_scan(_trajectory(1)[0], geom, clip=grid)
    far = np.array([[-140.0, -140.0]] * 4 + [[140.0, 140.0]] * 4) + np.tile([[0.01, 0.02], [0.05, 0.07], [0.1, 0.03], [0.12, 0.11]], (2, 1))
    ref = oracle.Frame((0, 0, 0), 300, 300, 0.25)
    new = oracle.Frame((0, 0, 0), 300, 300, 300.0)
    for q in scan.get():
        new.add_point(q[0], q[1])
    P, I = 12, 20
    cfg, ocfg = capi.PSOConfig.make(I, P), oracle.PSOConfig.make(I, P)
    table = oracle.glibc_rand(3, 3 + 3 * P + 6 * P * I)
    for step in range(2):   # the scan's cells alone (a box that fits LDS), then with the two far corners (one that does not)
        if step == 0:
            m.insert(scan, (0, 0, 0))
            ref.update((0, 0, 0), new)
        else:
            m.insert_host(far, (0, 0, 0))
            for q in far:
                ref.add_point(q[0], q[1])
        m.build()
        ref.build()
        info = m.info()
        assert info["status"] == 0
        for mode in (capi.SCORE_EXACT, capi.SCORE_F64):
            pose, cost, st = m.align(scan, (0.05, -0.03, 0.01), (0.5, 0.5, 0.2), cfg, rand_table=table, mode=mode)
            o_pose, o_cost, _ = ref.pso(np.array([0.05, -0.03, 0.01]), new, np.array([0.5, 0.5, 0.2]), ocfg, table=table)
            assert np.array_equal(pose, o_pose) and abs(cost - o_cost) <= 1e-11 * abs(o_cost), (step, mode, pose, o_pose)
        if step == 1:
            assert (info["x1"] - info["x0"] + 1) * (info["y1"] - info["y0"] + 1) > 1000000, info
    _compare_cells(m.cells(), ref.cells())
    with pytest.raises(capi.NdtpsoError) as e:
        capi.ResidentMap(ctx, capi.Grid(300, 300, 0.2))
    assert e.value.code == capi.E_ARG and "2^21" in str(e.value)


def test_resident_sequence_matches_golden_fixture(ctx):
    """Fixture G5 (tests/golden/oracle_golden_sequence.npz) through the device-resident path, no oracle at run time:
    poses of the 14-scan sequence, the cells' window state, the occupancy grid, the stored points, resetCells."""
    import os
    from ndtpso_slam_amd import capi
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden_sequence.npz"))
    frame, n_beams, n_scans, P, I, seed = (int(v) for v in g["params"])
    cs, ogcs = float(g["cell_side"]), float(g["og_cell_size"])
    geom = capi.ScanGeom(n_beams, float(g["angle_min"]), float(g["angle_inc"]), float(g["range_max"]), 0.1)
    grid = capi.Grid(frame, frame, cs)
    cfg = capi.PSOConfig.make(I, P)
    rmap = capi.ResidentMap(ctx, grid, og_cell_size=ogcs, pool_bytes=16 << 20)
    scan = capi.ResidentScan(ctx, 1024)
    prev, prev_pose, pose_diff, s_iter = np.zeros(3), np.zeros(3), np.zeros(3), 0
    stream = _glibc_rand(seed, (3 + 3 * P + 6 * P * I) * n_scans)
    poses = []
    for k in range(n_scans):
        scan.load_scan(g["ranges"][k], geom, clip=grid)
        if k == 0:
            pose = prev.copy()
        else:
            # NDTFrame::align, ndtframe.cpp:251-266; one srand(seed) stream runs on across the alignments
            dev = np.array([.1, .1, 3.1415e-3]) if s_iter < 2 else np.abs(2. * pose_diff)
            s_iter += 1
            n_draw = 3 + 3 * P + 6 * P * I
            table = stream[(k - 1) * n_draw:k * n_draw]
            pose, _, _ = rmap.align(scan, prev, dev, cfg, rand_table=table, mode=capi.SCORE_F64)
            pose_diff, prev_pose = pose - prev_pose, pose
        prev = pose
        rmap.insert(scan, pose)
        poses.append(pose)
    poses = np.array(poses)
    print("max |dpose| vs fixture", np.abs(poses - g["poses"]).max(axis=0))
    assert np.abs(poses - g["poses"]).max() < 1e-6
    rmap.build()
    cells = rmap.cells()
    assert np.array_equal([c["index"] for c in cells], g["cell_index"])
    assert np.array_equal([c["count"] for c in cells], g["cell_count"])
    assert np.array_equal([c["slot"] for c in cells], g["cell_slot"])
    built = np.array([c["built"] for c in cells])
    assert np.array_equal(built, g["cell_built"].astype(bool))
    mean = np.array([c["mean"] for c in cells])[built]
    icov = np.array([c["icov"] for c in cells])[built]
    assert np.allclose(mean, g["cell_mean"][built], rtol=0, atol=1e-6)
    assert np.allclose(icov, g["cell_icov"][built], rtol=1e-4, atol=1e-6)
    og, w, h, ext = rmap.occupancy()
    assert (w, h) == tuple(g["og_shape"]) and ext == tuple(int(v) for v in g["og_extent"])
    want = np.zeros(w * h, dtype=np.int8)
    want[g["og_nonzero_index"]] = g["og_nonzero_value"]
    assert np.abs(og.astype(int) - want.astype(int)).max() <= 1
    pts = rmap.points()
    assert len(pts) == int(g["points_count"]) and np.allclose(pts[:64], g["points_head"], atol=1e-6)
    assert np.allclose(pts.sum(axis=0), g["points_sum"], atol=1e-3)
    rmap.reset()
    for k in (0, 1):
        scan.load_scan(g["ranges"][k], geom, clip=None)
        rmap.insert(scan, None)
        rmap.build()
    cells = rmap.cells()
    assert np.array_equal([c["count"] for c in cells], g["reset_cell_count"])
    assert np.array_equal([c["built"] for c in cells], g["reset_cell_built"].astype(bool))


def _glibc_rand(seed, n):
    """glibc srand(seed); rand() x n (TYPE_3 additive feedback: r[i] = r[i-31] + r[i-3], output >> 1)"""
    r = [0] * (34 + 310 + n)
    r[0] = seed if seed else 1
    for i in range(1, 31):
        r[i] = (16807 * r[i - 1]) % 2147483647
    for i in range(31, 34):
        r[i] = r[i - 31]
    for i in range(34, 344 + n):
        r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
    return np.array([v >> 1 for v in r[344:344 + n]], dtype=np.int32)


@pytest.mark.parametrize("seed,frame_w,frame_h,cs,ogcs", [
    (1, 20, 12, 0.7, 0.2), (2, 12, 20, 1.0, 0.5), (3, 16, 16, 0.5, 0.0),
    # found by tests/campaigns/fuzz_campaign.py: a cell of three coincident points (NaN inverse covariance, NaN scores) --
    (445622, 30, 12, 0.5, 0.0),    # ... the NaN must survive the masking of the score loop
    (754063, 30, 30, 0.5, 0.1)])   # ... and a particle whose pbest is NaN never moves the gbest (core.cpp:94-104)
def test_resident_map_random_operation_sequences(ctx, oracle, seed, frame_w, frame_h, cs, ogcs):
    run_operation_sequence(ctx, oracle, seed, frame_w, frame_h, cs, ogcs)


def run_operation_sequence(ctx, oracle, seed, frame_w, frame_h, cs, ogcs):
    """Differential fuzz of the resident map against the oracle's NDTFrame: random sequences of addPoint batches
    (empty, single, several tiles, points on cell edges and frame borders), updates with a pose, builds, alignments
    (which build lazily), resetCells -- on non-square frames and a cell side that is not a power of two.  After every
    build the cells' observable state must be identical; at the end every stored point and the occupancy grid too."""
    from ndtpso_slam_amd import capi
    rng = np.random.default_rng(seed)
    grid = capi.Grid(frame_w, frame_h, cs)
    rmap = capi.ResidentMap(ctx, grid, og_cell_size=ogcs, pool_bytes=32 << 20)
    ref = oracle.Frame((0, 0, 0), frame_w, frame_h, cs)
    if ogcs > 0:
        ref.enable_occupancy_grid(ogcs)
    scan = capi.ResidentScan(ctx, 4096)
    hw, hh = frame_w / 2, frame_h / 2
    cfg, ocfg = capi.PSOConfig.make(6, 9), oracle.PSOConfig.make(6, 9)

    def cloud(n):
        xy = np.stack([rng.uniform(-hw * 1.1, hw * 1.1, n), rng.uniform(-hh * 1.1, hh * 1.1, n)], axis=1)
        if n:
            k = rng.integers(0, n, size=max(1, n // 9))
            xy[k] = np.round(xy[k] / cs) * cs                 # exactly on cell edges / frame borders
            xy[rng.integers(0, n, size=max(1, n // 5))] *= 0.2   # a dense centre: cells that rotate
        return xy

    empty = oracle.Frame((0, 0, 0), frame_w, frame_h, float(max(frame_w, frame_h)))
    n_builds = n_aligns = n_exact32 = 0
    for it in range(140):
        op = rng.choice(["add", "add", "update", "build", "align", "reset"], p=[.3, .2, .2, .15, .1, .05])
        if op in ("add", "update"):
            n = int(rng.choice([0, 1, 2, 7, 64, 300, 1024, 1025, 2500]))
            xy = cloud(n)
            pose = None if op == "add" else rng.uniform(-1, 1, 3) * (0.3, 0.3, 0.2)
            rmap.insert_host(xy, pose)
            if pose is not None and n:
                c, s = np_ref.cos_sin(pose[2])   # one sincos(), like the reference built by GCC
                xy = np.stack([xy[:, 0] * c - xy[:, 1] * s + pose[0], xy[:, 0] * s + xy[:, 1] * c + pose[1]], axis=1)
            for q in xy:
                ref.add_point(q[0], q[1])
            if pose is not None:     # update() clears `built` whether or not a point follows (ndtframe.cpp:188);
                ref.update((0, 0, 0), empty)   # addPoint only when the point lands in the frame (:219-224)
        elif op == "build":
            rmap.build()
            ref.build()
            n_builds += 1
            _compare_cells(rmap.cells(), ref.cells())
        elif op == "align":
            new_xy = cloud(200) * 0.5
            new = oracle.Frame((0, 0, 0), frame_w, frame_h, float(max(frame_w, frame_h)))
            for q in new_xy:
                new.add_point(q[0], q[1])
            new_xy = new.points()
            if len(new_xy) == 0:
                continue
            scan.set(new_xy)
            table = oracle.glibc_rand(int(rng.integers(1, 1 << 30)), 3 + 3 * 9 + 6 * 9 * 6)
            got, cost, _ = rmap.align(scan, (0, 0, 0), (.2, .2, .05), cfg, rand_table=table, mode=capi.SCORE_F64)
            want, want_cost, _ = ref.pso((0, 0, 0), new, (.2, .2, .05), ocfg, table=table)
            probes = rng.uniform(-1, 1, (5, 3)) * (.3, .3, .1)
            got_c = rmap.cost(scan, probes, mode=capi.SCORE_F64)            # cost_function against the resident map
            want_c = np.array([ref.cost(q, new) for q in probes])
            fin = np.isfinite(want_c)
            assert np.array_equal(np.isfinite(got_c), fin) and np.allclose(got_c[fin], want_c[fin], rtol=1e-9, atol=1e-9)
            # (after a resetCells the window's stale partial terms can make a covariance indefinite and a cost infinite --
            # in the reference too; equal infinities count as equal)
            assert np.abs(got - want).max() < 1e-9, (it, got, want)
            # the exact mode (the drop-in library's default) returns what the fp64 mode returns, NaN costs included
            gotx, costx, _ = rmap.align(scan, (0, 0, 0), (.2, .2, .05), cfg, rand_table=table, mode=capi.SCORE_EXACT)
            assert np.array_equal(gotx, got) and (costx == cost or (np.isnan(costx) and np.isnan(cost))), (it, gotx, got, costx, cost)
            got32, _, _ = rmap.align(scan, (0, 0, 0), (.2, .2, .05), cfg, rand_table=table, mode=capi.SCORE_F32)
            assert np.abs(got32 - want).max() < 1e-3, (it, got32, want)     # BASELINE tolerance of the fp32 score
            n_exact32 += int(np.array_equal(got32, want))
            # (a cell of coincident points has a NaN inverse covariance: NaN costs on both sides count as equal)
            assert (cost == want_cost or (np.isnan(cost) and np.isnan(want_cost))
                    or abs(cost - want_cost) < 1e-9 * max(1.0, abs(want_cost))), (it, cost, want_cost)
            n_aligns += 1
            _compare_cells(rmap.cells(), ref.cells())
        else:
            rmap.reset()
            ref.reset_cells()
    rmap.build()
    ref.build()
    _compare_cells(rmap.cells(), ref.cells())
    assert np.array_equal(rmap.points(), ref.points_all())
    if ogcs > 0:
        og, w, h, ext = rmap.occupancy()
        want, ww, wh, mm = ref.occupancy_grid()
        assert (w, h, ext) == (ww, wh, mm)
        d = np.abs(og.astype(int) - want.astype(int))
        assert d.max() <= 1 and (d > 0).sum() <= max(2, 0.01 * (want != 0).sum())
    assert rmap.info()["status"] & 1 == 0 and n_builds > 5 and n_aligns > 1  # (a sequence of 140 draws has 14 alignments on average; 3 happen)
    return n_aligns, n_exact32


def test_speculative_build_is_invisible(ctx, oracle):
    """ndtpso_map_speculate_build builds cells and table ahead of the alignment that will want them.  Whatever comes
    next -- the alignment, an explicit build, another insert (the build is taken back), a reset, an export -- the map
    behaves exactly as the lazily built reference frame."""
    from ndtpso_slam_amd import capi
    rng = np.random.default_rng(21)
    cs = 1.0
    grid = capi.Grid(16, 12, cs)
    rmap = capi.ResidentMap(ctx, grid, og_cell_size=0.25, pool_bytes=16 << 20)
    ref = oracle.Frame((0, 0, 0), 16, 12, cs)
    ref.enable_occupancy_grid(0.25)
    scan = capi.ResidentScan(ctx, 1024)
    cfg, ocfg = capi.PSOConfig.make(5, 8), oracle.PSOConfig.make(5, 8)

    def add(n):
        xy = np.stack([rng.uniform(-8.5, 8.5, n), rng.uniform(-6.5, 6.5, n)], axis=1)
        xy[: n // 3] *= 0.15
        rmap.insert_host(xy)
        for q in xy:
            ref.add_point(q[0], q[1])

    def check(points=True):
        _compare_cells(rmap.cells(), ref.cells())
        if points:
            assert np.array_equal(rmap.points(), ref.points_all())
        og, w, h, ext = rmap.occupancy()
        want, ww, wh, mm = ref.occupancy_grid()
        assert (w, h, ext) == (ww, wh, mm) and np.abs(og.astype(int) - want.astype(int)).max() <= 1

    n_aligned = 0
    for it in range(40):
        add(int(rng.integers(50, 600)))
        rmap.speculate_build()
        nxt = rng.choice(["align", "build", "insert", "export", "reset", "twice"])
        if nxt == "align":
            new_xy = np.stack([rng.uniform(-3, 3, 150), rng.uniform(-3, 3, 150)], axis=1)
            new = oracle.Frame((0, 0, 0), 16, 12, 16.0)
            for q in new_xy:
                new.add_point(q[0], q[1])
            scan.set(new.points())
            table = oracle.glibc_rand(int(rng.integers(1, 1 << 30)), 3 + 3 * 8 + 6 * 8 * 5)
            # (exact mode: the alignment is enqueued before the host has seen the speculative build's header and binds
            # the table's window on the device -- same pose and cost, bit for bit)
            mode = capi.SCORE_F64 if n_aligned % 2 else capi.SCORE_EXACT
            n_aligned += 1
            got, cost, _ = rmap.align(scan, (0, 0, 0), (.2, .2, .05), cfg, rand_table=table, mode=mode)
            want, want_cost, _ = ref.pso((0, 0, 0), new, (.2, .2, .05), ocfg, table=table)
            assert np.array_equal(got, want) and (cost == want_cost or abs(cost - want_cost) < 1e-9 * max(1, abs(want_cost)))
            check()
        elif nxt == "build":
            rmap.build()
            ref.build()
            check()
        elif nxt == "insert":
            pass                              # the next round's insert arrives while the speculation is pending
        elif nxt == "export":
            check()                           # the unbuilt state must be what is exported
            rmap.build()
            ref.build()
            check()
        elif nxt == "reset":
            rmap.reset()
            ref.reset_cells()
            check()
        else:
            rmap.speculate_build()            # idempotent
            rmap.build()
            rmap.speculate_build()            # nothing to do: already built
            ref.build()
            check()
    rmap.build()
    ref.build()
    check()
