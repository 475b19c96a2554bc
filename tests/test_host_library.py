"""The C++ drop-in (host/libndtpso_slam.so): exports on CPU; on the GPU the ROS-free node replay
(ndtpso_slam_node.cpp:177-244 call sequence, accumulated map with sliding-window cells) against the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

import np_ref

from conftest import FRAME_M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")


def _build():
    from ndtpso_slam_amd import build as hip_build
    hip_build.build_hip()
    subprocess.check_call(["make", "-C", HOST, "-s"])


def test_host_library_exports_reference_api():
    _build()
    out = subprocess.check_output(["nm", "-D", "--demangle", os.path.join(HOST, "libndtpso_slam.so")], text=True)
    for sym in ["NDTFrame::NDTFrame(", "NDTFrame::loadLaser(", "NDTFrame::align(", "NDTFrame::update(",
                "NDTFrame::build()", "NDTFrame::addPoint(", "NDTFrame::getCellIndex(", "NDTFrame::addPose(",
                "NDTFrame::dumpMap(", "NDTFrame::resetCells()", "NDTFrame::transform(", "NDTCell::addPoint(",
                "NDTCell::build()", "NDTCell::normalDistribution(", "NDTCell::reset()", "pso_optimization(",
                "cost_function("]:
        assert sym in out, sym
    # the host library carries no device code and no CPU restatement of the hot path
    needed = subprocess.check_output(["readelf", "-d", os.path.join(HOST, "libndtpso_slam.so")], text=True)
    assert "libndtpso_hip.so" in needed


def test_abi_tag_records_the_vector_type_choice():
    """ndtpso_slam/linalg.h: the library defines ndtpso_slam_abi_with_eigen or ndtpso_slam_abi_without_eigen for the
    vector types IT was built with, and every consumer references the symbol of its own choice -- a node compiled with
    Eigen against a library compiled without (different NDTCell layout) fails at link time instead of at run time."""
    _build()
    out = subprocess.check_output(["nm", "-D", os.path.join(HOST, "libndtpso_slam.so")], text=True)
    tags = [l.split()[-1] for l in out.splitlines() if "ndtpso_slam_abi_" in l and " U " not in l]
    assert len(tags) == 1 and tags[0] in ("ndtpso_slam_abi_with_eigen", "ndtpso_slam_abi_without_eigen"), tags
    other = "ndtpso_slam_abi_with_eigen" if tags[0].endswith("without_eigen") else "ndtpso_slam_abi_without_eigen"
    # the consumers built here reference the library's tag; one that was compiled for the other choice does not link
    und = subprocess.check_output(["nm", "-D", os.path.join(HOST, "replay", "node_api_compile")], text=True)
    assert tags[0] in und
    src = 'extern "C" int %s; int main() { return %s; }' % (other, other)
    r = subprocess.run(["g++", "-x", "c++", "-", "-o", "/dev/null", "-L" + HOST, "-lndtpso_slam",
                        "-L" + os.path.join(ROOT, "ndtpso_slam_amd", "lib"), "-lndtpso_hip"], input=src, text=True,
                       capture_output=True)
    assert r.returncode != 0 and other in r.stderr


def test_node_call_sequence_links_and_survives_a_missing_device(tmp_path):
    """host/replay/node_api_compile.cpp makes exactly the calls ndtpso_slam_node.cpp makes (:64-78, 110, 155, 167, 186,
    194, 198, 202, 206, 229-230).  Without arguments: link check.  On a device that does not exist every device call
    fails: nothing aborts, align() returns its initial guess, the failures are counted and the text is available
    (ndtpso_slam/status.h) -- and nothing is computed on the CPU instead."""
    _build()
    exe = os.path.join(HOST, "replay", "node_api_compile")
    out = subprocess.check_output([exe], text=True)
    assert out.startswith("NDT_WINDOW_SIZE 100 PSO 30 x 50")
    env = {k: v for k, v in os.environ.items() if k != "NDTPSO_ABORT_ON_ERROR"}
    r = subprocess.run([exe, str(tmp_path)], text=True, capture_output=True, env=dict(env, NDTPSO_DEVICE="99"))
    assert r.returncode == 1, (r.stdout, r.stderr)                      # errors were counted, the process lived
    last = r.stdout.strip().splitlines()[-1].split()
    assert last[0] == "pose" and [float(v) for v in last[1:4]] == [0.0, 0.0, 0.0] and int(last[5]) > 0
    assert "align failed" in r.stderr and "the call is skipped" in r.stderr
    assert r.stderr.count("align failed") == 1                          # logged once per call site
    # the node update()s the map at whatever pose align() returned (ndtpso_slam_node.cpp:194-198): after a failed align that
    # is the unrefined guess, and the merge is refused rather than allowed to corrupt the map
    assert "update after a failed align (scan not merged)" in r.stderr
    assert os.path.exists(str(tmp_path / "node_api.pose.csv"))         # the shutdown export still ran
    # NDTPSO_ABORT_ON_ERROR=1: the old behaviour for tests and debugging
    r = subprocess.run([exe, str(tmp_path)], text=True, capture_output=True,
                       env=dict(os.environ, NDTPSO_DEVICE="99", NDTPSO_ABORT_ON_ERROR="1"))
    assert r.returncode < 0


def test_bulk_rand_draw_is_indistinguishable_from_rand():
    """host/replay/rand_check.cpp: the bulk draw of the std::rand() stream (glibc state advanced in place) gives the
    outputs of n rand() calls and leaves the generator where they would; also with the slow path forced."""
    _build()
    exe = os.path.join(HOST, "replay", "rand_check")
    out = subprocess.check_output([exe], text=True)
    assert out.startswith("ok:"), out
    out = subprocess.check_output([exe], text=True, env=dict(os.environ, NDTPSO_SLOW_RAND="1"))
    assert out.startswith("ok:"), out


def _trajectory(n_scans, seed=4):
    from ndtpso_slam_amd import synth
    rng = np.random.default_rng(seed)
    s = np.linspace(0.0, 0.6, n_scans)
    poses = np.stack([2.0 + 1.2 * s, -1.0 + 0.8 * np.sin(1.5 * s), 0.3 + 0.25 * s], axis=1)   # world poses
    clean = synth.raycast(poses)
    r = np.where(clean > 0, clean + rng.normal(0, 0.01, clean.shape), 0.0).astype(np.float32)
    return r, poses


@pytest.mark.gpu
@pytest.mark.parametrize("resident", ["1", "0"])
def test_node_replay_matches_oracle(tmp_path, oracle, resident, monkeypatch):
    """20 scans through loadLaser -> align -> update with the default 30 x 50 PSO (what the node really runs):
    the accumulated map exercises the sliding window (cells pass 50 points and open new slots) and the
    |2*pose_diff| deviation rule; one srand() at start, the stream runs on across alignments."""
    from ndtpso_slam_amd import synth
    _build()
    # resident: map, windows and scans stay on the GPU (ndtpso_map_*); 0: the frame's bookkeeping on the host
    monkeypatch.setenv("NDTPSO_RESIDENT", resident)
    n_scans, P, I, seed, cs = 20, 30, 50, 7, 0.5
    ranges, _ = _trajectory(n_scans)
    path = tmp_path / "scans.bin"
    with open(path, "wb") as f:
        np.array([n_scans, synth.N_BEAMS], dtype=np.int32).tofile(f)
        np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
        ranges.tofile(f)
    out = subprocess.check_output([os.path.join(HOST, "replay", "node_replay"), str(path), str(FRAME_M), str(cs),
                                   str(I), str(P), str(seed)], text=True, env=dict(os.environ, NDTPSO_SCORE="f64"))
    got = np.array([[float(v) for v in line.split()[1:]] for line in out.strip().splitlines()])
    assert got.shape == (n_scans, 3)

    cfg = oracle.PSOConfig.make(I, P)
    n_draw = 3 + 3 * P + 6 * P * I
    stream = oracle.glibc_rand(seed, n_draw * n_scans)
    ref = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, cs)
    cur = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, cs)
    prev = np.zeros(3)
    want = []
    k_align = 0
    for k in range(n_scans):
        cur.load_laser(ranges[k], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
        if k == 0:
            pose = prev.copy()
        else:
            pose = ref.align(prev, cur, cfg, table=stream[k_align * n_draw:(k_align + 1) * n_draw])
            k_align += 1
        prev = pose
        ref.update(pose, cur)
        want.append(pose)
        cur = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
    want = np.array(want)
    d = np.abs(got - want)
    print("node replay max |dpose|", d.max(axis=0))
    assert d[:, :2].max() < 1e-3 and d[:, 2].max() < 1e-3
    # the fp32 score mode follows the same trajectory
    out32 = subprocess.check_output([os.path.join(HOST, "replay", "node_replay"), str(path), str(FRAME_M), str(cs),
                                     str(I), str(P), str(seed)], text=True, env=dict(os.environ, NDTPSO_SCORE="f32"))
    got32 = np.array([[float(v) for v in line.split()[1:]] for line in out32.strip().splitlines()])
    d32 = np.abs(got32 - want)
    print("node replay (fp32 score) max |dpose|", d32.max(axis=0))
    assert d32[:, :2].max() < 1e-3 and d32[:, 2].max() < 1e-3
    # the library's default score mode (exact: fp32 score, undecidable comparisons arbitrated in fp64) prints what the
    # fp64 score mode prints, digit for digit
    env_default = {k: v for k, v in os.environ.items() if k != "NDTPSO_SCORE"}
    out_x = subprocess.check_output([os.path.join(HOST, "replay", "node_replay"), str(path), str(FRAME_M), str(cs),
                                     str(I), str(P), str(seed)], text=True, env=env_default)
    assert out_x == out
    if resident == "1":
        # late window binding (ndtpso_map_align): by default the alignment is enqueued before the host has seen the
        # table's header and binds its window on the device; waiting for the header instead, or a bound the table
        # exceeds (every alignment flagged by the kernel and redone the usual way), print the same digits
        exe = [os.path.join(HOST, "replay", "node_replay"), str(path), str(FRAME_M), str(cs), str(I), str(P), str(seed)]
        r = subprocess.run(exe, text=True, capture_output=True, check=True, env=dict(env_default, NDTPSO_LOG_LATE="1"))
        m = re.search(r"late window binding: (\d+) alignments, (\d+) redone", r.stderr)
        assert m and int(m.group(1)) >= n_scans - 3 and int(m.group(2)) == 0, r.stderr
        assert r.stdout == out
        r = subprocess.run(exe, text=True, capture_output=True, check=True, env=dict(env_default, NDTPSO_LOG_LATE="1", NDTPSO_LATE_WINDOW="0"))
        assert "late window binding" not in r.stderr and r.stdout == out
        r = subprocess.run(exe, text=True, capture_output=True, check=True,
                           env=dict(env_default, NDTPSO_LOG_LATE="1", NDTPSO_LATE_TEST_OVERFLOW="1"))
        m = re.search(r"late window binding: (\d+) alignments, (\d+) redone", r.stderr)
        assert m and int(m.group(1)) == 0 and int(m.group(2)) >= n_scans - 3, r.stderr
        assert r.stdout == out
    # like the reference (ndtframe.cpp:257) align() runs 30 x 50 whatever PSO configuration the frame was given
    out_cfg = subprocess.check_output([os.path.join(HOST, "replay", "node_replay"), str(path), str(FRAME_M), str(cs),
                                       "7", "11", str(seed)], text=True, env=dict(os.environ, NDTPSO_SCORE="f64"))
    assert out_cfg == out


@pytest.mark.gpu
@pytest.mark.parametrize("resident", ["1", "0"])
def test_node_replay_map_export(tmp_path, oracle, resident, monkeypatch):
    """the shutdown export of the node (ndtpso_slam_node.cpp:141-172 -> NDTFrame::dumpMap,
    ndtframe.cpp:268-422) and the occupancy grid of the reference frame (ndtframe.cpp:79-112) after a replay."""
    import struct
    import zlib
    from ndtpso_slam_amd import synth
    _build()
    monkeypatch.setenv("NDTPSO_RESIDENT", resident)
    n_scans, P, I, seed, cs, ogcs = 8, 20, 20, 3, 0.5, 0.1
    SAVE_EACH = 3   # the node's SAVE_DATA_TO_FILE_EACH_NUM_ITERS is 10; 3 puts three merges into an 8-scan replay
    ranges, _ = _trajectory(n_scans)
    path = tmp_path / "scans.bin"
    with open(path, "wb") as f:
        np.array([n_scans, synth.N_BEAMS], dtype=np.int32).tofile(f)
        np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
        ranges.tofile(f)
    prefix = str(tmp_path / "run")
    out = subprocess.check_output([os.path.join(HOST, "replay", "node_replay"), str(path), str(FRAME_M), str(cs), str(I),
                                   str(P), str(seed), str(ogcs), prefix, "20"], text=True,
                                  env=dict(os.environ, NDTPSO_SCORE="f64", NDTPSO_ALIGN_FRAME_CONFIG="1",
                                           NODE_REPLAY_SAVE_EACH=str(SAVE_EACH)))
    got = np.array([[float(v) for v in line.split()[1:]] for line in out.strip().splitlines()])

    # the same run on the oracle (a short 20 x 20 PSO: NDTPSO_ALIGN_FRAME_CONFIG lets align() use the frame's config)
    cfg = oracle.PSOConfig.make(I, P)
    n_draw = 3 + 3 * P + 6 * P * I
    stream = oracle.glibc_rand(seed, n_draw * n_scans)
    ref = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, cs)
    ref.enable_occupancy_grid(ogcs)
    cur = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, cs)
    prev = np.zeros(3)
    n_map_points = 0
    for k in range(n_scans):
        cur.load_laser(ranges[k], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
        pose = prev.copy() if k == 0 else ref.align(prev, cur, cfg, table=stream[(k - 1) * n_draw:k * n_draw])
        prev = pose
        ref.update(pose, cur)
        # what the one-cell global map keeps: the node merges every SAVE_EACH-th scan, starting with the first
        # (ndtpso_slam_node.cpp:200-205), and of it the points that fall strictly inside the frame
        if k % SAVE_EACH == 0:
            c, s = np_ref.cos_sin(pose[2])   # one sincos(), like the reference built by GCC
            pts = cur.points()
            gx, gy = pts[:, 0] * c - pts[:, 1] * s + pose[0], pts[:, 0] * s + pts[:, 1] * c + pose[1]
            n_map_points += int(((np.abs(gx) < FRAME_M / 2) & (np.abs(gy) < FRAME_M / 2)).sum())
        cur = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
    assert np.abs(got[-1] - prev).max() < 1e-6

    # pose.csv / map.csv / gnuplot: the reference's text formats
    lines = open(prefix + ".pose.csv").read().splitlines()
    assert lines[0] == "timestamp,xP,yP,thP,xO,yO,thO" and len(lines) == 1 + n_scans
    for k, row in enumerate(lines[1:]):
        assert row == "%.6f,%.5f,%.5f,%.5f" % (0.025 * k, got[k, 0], got[k, 1], got[k, 2])
    lines = open(prefix + ".map.csv").read().splitlines()
    assert lines[0] == "x,y" and len(lines) == 1 + n_map_points
    gp = open(prefix + ".gnuplot").read()
    assert gp.startswith("set datafile separator ','\nset key autotitle columnhead\nset size ratio -1\nplot '")
    assert gp.endswith("\npause 1000\n") and "using 2:3 title 'Pose (LiDAR)'" in gp
    assert not os.path.exists(prefix + "-ref-frame.pose.csv")          # save_poses = false for the reference frame

    # occupancy grid of the reference frame vs the oracle's
    raw = open(prefix + "-ref-frame.og.bin", "rb").read()
    w, h, x0, x1, y0, y1 = struct.unpack("6I", raw[:24])
    og = np.frombuffer(raw[24:], dtype=np.int8)
    want, ww, wh, mm = ref.occupancy_grid()
    assert (w, h) == (ww, wh) and (x0, x1, y0, y1) == mm
    diff = np.abs(og.astype(int) - want.astype(int))
    print("occupancy grid: %d cells > 0, %d differ, max |d| %d" % ((want > 0).sum(), (diff > 0).sum(), diff.max()))
    assert diff.max() <= 1 and (diff > 0).sum() <= 0.01 * max(1, (want != 0).sum())

    # the two images are valid PNGs of the reference's sizes
    def png(path):
        b = open(path, "rb").read()
        assert b[:8] == b"\x89PNG\r\n\x1a\n"
        cols, rows, depth, ctype = struct.unpack(">IIBB", b[16:26])
        pos, idat = 8, b""
        while pos < len(b):
            n, tag = struct.unpack(">I4s", b[pos:pos + 8])
            assert zlib.crc32(b[pos + 4:pos + 8 + n]) == struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
            if tag == b"IDAT":
                idat += b[pos + 8:pos + 8 + n]
            pos += 12 + n
        px = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(rows, -1)[:, 1:]
        return rows, cols, ctype, px
    rows, cols, ctype, px = png("%s-w100-30p50i-%dx%d-c%.2f-20ppm.png" % (prefix, FRAME_M, FRAME_M, float(FRAME_M)))
    assert (rows, cols, ctype) == (FRAME_M * 20, FRAME_M * 20, 2) and (px == 0).any()
    rows, cols, ctype, px = png("%s-ref-frame-%dx%d-cell%.2fm-occupancy-grid.png" % (prefix, w, h, ogcs))
    assert (rows, cols, ctype) == (y1 - y0 + 1, x1 - x0 + 1, 0)
    # pixel = 255 - 2.55 og, rows flipped, cropped to the extent
    img = want.reshape(h, w)[y0:y1 + 1, x0:x1 + 1][::-1]      # og[x + height*y] with height == width here
    expect = np.where(img > 0, (255.0 - img * 2.55).astype(np.uint8), 255)
    assert (np.abs(px.astype(int) - expect.astype(int)) <= 3).all()


@pytest.mark.gpu
def test_frame_api_resident_and_host_frames_agree(tmp_path, oracle):
    """The public NDTFrame / core.h API beyond the node's calls (host/replay/frame_api_check.cpp): frames whose state
    lives on the GPU and frames kept by the host print the same numbers, and the first steps match the oracle."""
    from ndtpso_slam_amd import synth
    _build()
    ranges, _ = _trajectory(3)
    path = tmp_path / "scans.bin"
    with open(path, "wb") as f:
        np.array([3, synth.N_BEAMS], dtype=np.int32).tofile(f)
        np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
        ranges.tofile(f)
    outs = {}
    for resident in ("1", "0"):
        prefix = str(tmp_path / ("dump" + resident))
        outs[resident] = subprocess.check_output([os.path.join(HOST, "replay", "frame_api_check"), str(path), prefix],
                                                 text=True, env=dict(os.environ, NDTPSO_RESIDENT=resident,
                                                                     NDTPSO_SCORE="f64"))
    print(outs["1"])
    assert outs["1"] == outs["0"]
    # NDTFrame::addScan (north star's name for loadLaser + update): the frame it fills equals the two-call one
    add = [l for l in outs["1"].splitlines() if l.startswith("addScan")]
    assert len(add) == 2 and add[0] == add[1] and " built 0 " not in add[0]
    assert outs["1"].strip().splitlines()[-1] == "errors 0"
    for ext in (".pose.csv", ".map.csv"):
        assert open(str(tmp_path / "dump1") + ext).read() == open(str(tmp_path / "dump0") + ext).read()
    lines = {l.split()[0]: l.split()[1:] for l in outs["1"].splitlines() if l.split()[0] in ("cost", "pso")}
    a = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, 0.5)
    a.load_laser(ranges[0], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
    a.build()
    b = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
    b.load_laser(ranges[1], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
    want_cost = a.cost((0.05, -0.03, 0.01), b)
    assert abs(float(lines["cost"][0]) - want_cost) <= 1e-9 * abs(want_cost)
    pose, _, _ = a.pso((0, 0, 0), b, (.1, .1, 3.1415e-3), oracle.PSOConfig.make(25, 20), seed=5)
    assert np.abs(np.array([float(v) for v in lines["pso"]]) - pose).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("resident", ["1", "0"])
def test_node_replay_through_window_wrap_around(tmp_path, oracle, resident):
    """A long replay on coarse cells: the busiest cells rotate through all 100 window slots and start reusing them
    (stale points of the previous lap still feed the covariance, the running sum restarts at zero, ndtcell.cpp:22-27,
    49, 61-65).  Both kinds of frames must follow the oracle pose by pose."""
    from ndtpso_slam_amd import synth
    _build()
    n_scans, P, I, seed, cs = 420, 8, 6, 5, 2.0
    s = np.linspace(0.0, 1.0, n_scans)
    poses = np.stack([4.0 * np.cos(2 * np.pi * s), 3.0 * np.sin(2 * np.pi * s), 2 * np.pi * s + np.pi / 2], axis=1)
    rng = np.random.default_rng(6)
    clean = synth.raycast(poses)
    ranges = np.where(clean > 0, clean + rng.normal(0, 0.01, clean.shape), 0.0).astype(np.float32)
    path = tmp_path / "scans.bin"
    with open(path, "wb") as f:
        np.array([n_scans, synth.N_BEAMS], dtype=np.int32).tofile(f)
        np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
        ranges.tofile(f)
    out = subprocess.check_output([os.path.join(HOST, "replay", "node_replay"), str(path), str(FRAME_M), str(cs), str(I),
                                   str(P), str(seed)], text=True,
                                  env=dict(os.environ, NDTPSO_SCORE="f64", NDTPSO_ALIGN_FRAME_CONFIG="1",
                                           NDTPSO_RESIDENT=resident))
    got = np.array([[float(v) for v in line.split()[1:]] for line in out.strip().splitlines()])
    cfg = oracle.PSOConfig.make(I, P)
    n_draw = 3 + 3 * P + 6 * P * I
    stream = oracle.glibc_rand(seed, n_draw * n_scans)
    ref = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, cs)
    prev = np.zeros(3)
    want = []
    n_inserted = 0
    for k in range(n_scans):
        cur = oracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
        cur.load_laser(ranges[k], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
        pose = prev.copy() if k == 0 else ref.align(prev, cur, cfg, table=stream[(k - 1) * n_draw:k * n_draw])
        prev = pose
        ref.update(pose, cur)
        n_inserted += len(cur.points())        # (all inside the frame here)
        want.append(pose)
    want = np.array(want)
    laps = max(c["slot"] for c in ref.cells())
    ref.build()
    assert len(ref.points_all()) < n_inserted      # reused slots dropped the points of their previous lap
    d = np.abs(got - want)
    first = np.nonzero(d.max(axis=1) > 0)[0]
    print("window wrap replay: max |dpose| %.3e, first differing scan %s, busiest cell at slot %d" % (d.max(), first[:1], laps))
    assert d.max() < 1e-9


def run_host_operation_sequence(oracle, workdir, seed, frame_w, frame_h, cs, ogcs, n_ops=60):
    """Differential fuzz of the C++ drop-in (host/replay/frame_fuzz.cpp drives the public NDTFrame / core.h API from a
    script) against the oracle: random addPoint batches (cell edges, frame borders, a dense centre whose cells rotate
    their windows, coincident points), update with a pose, build, resetCells, cost_function, pso_optimization and
    NDTFrame::align with its deviation rule.  Frames resident on the GPU and frames kept by the host, fp64 and fp32
    score.  Returns (alignments compared, fp32-score alignments that were bit-identical)."""
    rng = np.random.default_rng(seed)
    hw, hh = frame_w / 2, frame_h / 2
    ref = oracle.Frame((0, 0, 0), frame_w, frame_h, cs)
    if ogcs > 0:
        ref.enable_occupancy_grid(ogcs)
    hx = float.hex
    script = [f"frame {frame_w} {frame_h} {hx(float(cs))} {hx(float(ogcs))}"]
    expect = [("frame",) + ref.dims()]

    def cloud(n):
        xy = np.stack([rng.uniform(-hw * 1.1, hw * 1.1, n), rng.uniform(-hh * 1.1, hh * 1.1, n)], axis=1)
        if n:
            k = rng.integers(0, n, size=max(1, n // 9))
            xy[k] = np.round(xy[k] / cs) * cs                     # exactly on cell edges / frame borders
            xy[rng.integers(0, n, size=max(1, n // 5))] *= 0.2    # a dense centre: cells that rotate
            if n >= 7 and rng.random() < 0.3:
                xy[:3] = xy[0]                                    # three coincident points (NaN inverse covariance)
        return xy

    def pts(xy):
        return f"{len(xy)} " + " ".join(hx(float(v)) for v in xy.reshape(-1))

    def new_frame(xy):
        nf = oracle.Frame((0, 0, 0), frame_w, frame_h, float(max(frame_w, frame_h)))
        for q in xy:
            nf.add_point(q[0], q[1])
        return nf

    for _ in range(n_ops):
        op = rng.choice(["add", "update", "build", "reset", "cost", "pso", "align", "points"],
                        p=[.3, .2, .12, .04, .1, .08, .1, .06])
        if op == "add":
            xy = cloud(int(rng.choice([0, 1, 2, 7, 64, 300, 1100])))
            script.append("add " + pts(xy))
            for q in xy:
                ref.add_point(q[0], q[1])
            expect.append(("add", len(xy)))
        elif op == "update":
            xy = cloud(int(rng.choice([0, 3, 64, 300, 1100])))
            pose = rng.uniform(-1, 1, 3) * (0.3, 0.3, 0.2)
            script.append("update " + " ".join(hx(float(v)) for v in pose) + " " + pts(xy))
            ref.update(pose, new_frame(xy))
            expect.append(("update",))
        elif op == "build":
            script.append("build")
            ref.build()
            expect.append(("build", [(c["index"], c["built"], c["mean"]) for c in ref.cells()]))
        elif op == "reset":
            script.append("reset")
            ref.reset_cells()
            expect.append(("reset",))
        elif op == "points":
            script.append("points")
            xy = ref.points()
            sx = sy = 0.0
            for q in xy:
                sx += q[0]
                sy += q[1]
            expect.append(("points", len(xy), sx, sy))
        else:
            xy = cloud(200) * 0.5
            nf = new_frame(xy)
            if op == "cost":
                pose = rng.uniform(-1, 1, 3) * (.3, .3, .1)
                script.append("cost " + " ".join(hx(float(v)) for v in pose) + " " + pts(xy))
                expect.append(("cost", ref.cost(pose, nf), len(nf.points())))
            elif op == "pso":
                sd = int(rng.integers(1, 1 << 30))
                g, d = rng.uniform(-1, 1, 3) * (.1, .1, .02), (.2, .2, .05)
                I, P = int(rng.integers(0, 9)), int(rng.integers(1, 25))
                script.append(f"pso {sd} " + " ".join(hx(float(v)) for v in (*g, *d)) + f" {I} {P} " + pts(xy))
                pose, _, _ = ref.pso(g, nf, d, oracle.PSOConfig.make(I, P), seed=sd)
                expect.append(("pso", pose))
            else:
                sd = int(rng.integers(1, 1 << 30))
                g = rng.uniform(-1, 1, 3) * (.1, .1, .02)
                script.append(f"align {sd} " + " ".join(hx(float(v)) for v in g) + " " + pts(xy))
                expect.append(("align", ref.align(g, nf, None, seed=sd)))
    script.append("end")
    path = os.path.join(str(workdir), f"ops_{seed}.txt")
    with open(path, "w") as f:
        f.write("\n".join(script) + "\n")

    n_aligns = n_exact32 = 0
    for resident in ("1", "0"):
        for score in ("f64", "f32", "exact"):
            r = subprocess.run([os.path.join(HOST, "replay", "frame_fuzz"), path], text=True, capture_output=True,
                               env=dict(os.environ, NDTPSO_RESIDENT=resident, NDTPSO_SCORE=score))
            assert r.returncode == 0, (seed, resident, score, r.returncode, r.stderr[-2000:], r.stdout[-500:])
            out = r.stdout
            lines = [l.split() for l in out.splitlines()]
            assert len(lines) == len(expect), (resident, score, len(lines), len(expect))
            for k, (got, want) in enumerate(zip(lines, expect)):
                tag = (seed, resident, score, k, want[0])
                assert got[0] == want[0], tag
                if want[0] == "frame":
                    assert (int(got[1]), int(got[2])) == want[1:], tag
                elif want[0] == "build":
                    cells = want[1]
                    assert int(got[1]) == len(cells), tag
                    for i, (index, built, mean) in enumerate(cells):
                        gi, gb, gx, gy = got[2 + 4 * i: 6 + 4 * i]
                        assert (int(gi), bool(int(gb))) == (index, built), tag + (i,)
                        if built:
                            assert np.array_equal([float.fromhex(gx), float.fromhex(gy)], mean, equal_nan=True), \
                                tag + (i, index, float.fromhex(gx), float.fromhex(gy), mean)
                elif want[0] == "points":
                    assert int(got[1]) == want[1], tag
                    assert (float.fromhex(got[2]), float.fromhex(got[3])) == (want[2], want[3]), tag
                elif want[0] == "cost":
                    g, w = float.fromhex(got[1]), want[1]
                    tol = 1e-9 * max(1.0, abs(w)) if score != "f32" else max(1e-4 * max(1, want[2]), 1e-6 * abs(w))
                    assert (np.isnan(g) and np.isnan(w)) or g == w or abs(g - w) <= tol, tag + (g, w)
                elif want[0] in ("pso", "align"):
                    g = np.array([float.fromhex(v) for v in got[1:4]])
                    assert np.abs(g - want[1]).max() < (1e-9 if score != "f32" else 1e-3), tag + (g, want[1])
                    if score == "f32":
                        n_aligns += 1
                        n_exact32 += int(np.array_equal(g, want[1]))
    return n_aligns, n_exact32


@pytest.mark.gpu
@pytest.mark.parametrize("seed,frame_w,frame_h,cs,ogcs", [(11, 20, 12, 0.7, 0.2), (12, 16, 16, 0.5, 0.0), (13, 12, 30, 1.0, 0.5)])
def test_frame_api_random_operation_sequences(tmp_path, oracle, seed, frame_w, frame_h, cs, ogcs):
    _build()
    n_aligns, _ = run_host_operation_sequence(oracle, tmp_path, seed, frame_w, frame_h, cs, ogcs)
    assert n_aligns > 0


@pytest.mark.gpu
@pytest.mark.parametrize("R", [4, 32])
def test_replicas_on_threads_of_one_process_reproduce_their_single_process_logs(tmp_path, R):
    """SURVEY 8(b) "one context per host thread", 8(e) "replicas only": host/replay/node_replicas.cpp runs R node sequences on
    R host threads of ONE process -- the library gives each thread a device context and stream of its own, and
    ndtpso_slam_thread_srand a random stream of its own -- and replica r's pose log must be, byte for byte, what
    `node_replay ... seed + r` prints when it has the process to itself.  R = 32: more replicas than the device has hardware
    queues to run side by side -- alignments beyond sixteen in flight stay on one workgroup, waiting threads sleep between looks
    (round 6) --, same logs, no alignment left to the bounded wait."""
    from ndtpso_slam_amd import synth
    _build()
    n_scans, P, I, seed, cs = 16, 30, 50, 11, 0.5
    ranges, _ = _trajectory(n_scans, seed=9)
    path = tmp_path / "scans.bin"
    with open(path, "wb") as f:
        np.array([n_scans, synth.N_BEAMS], dtype=np.int32).tofile(f)
        np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
        ranges.tofile(f)
    alone = []
    for r in range(R):
        alone.append(subprocess.check_output([os.path.join(HOST, "replay", "node_replay"), str(path), str(FRAME_M), str(cs), str(I), str(P),
                                              str(seed + r)], text=True))
    assert len(set(alone)) == R                      # different seeds do give different trajectories
    prefix = str(tmp_path / "rep")
    import json
    for rep in range(2):                             # twice: the interleaving differs, the logs must not
        out = subprocess.check_output([os.path.join(HOST, "replay", "node_replicas"), str(path), str(FRAME_M), str(cs), str(I), str(P),
                                       str(seed), str(R), prefix], text=True)
        info = json.loads(out.strip().splitlines()[-1])
        assert info["replicas"] == R and info["failed_alignments"] == 0 and info["device_errors"] == 0, info
        # (a cluster that ran into the bounded wait is redone on one workgroup: same pose, a latency spike -- none in any run of the
        # round's library, but a busy box must not fail the suite over one)
        print("replicas %d: cluster timeouts %d, %.0f scans/s" % (R, info["cluster_timeouts"], info["aggregate_scans_per_s"]))
        assert info["cluster_timeouts"] <= (0 if R <= 16 else 4), info
        for r in range(R):
            with open("%s.%d.poses" % (prefix, r)) as f:
                assert f.read() == alone[r], "replica %d's log differs from its single-process log" % r


@pytest.mark.gpu
def test_short_lived_threads_recycle_their_device_contexts():
    """host/replay/thread_churn.cpp: forty host threads one after the other, each with a frame of its own aligned against a
    frame of the main thread's.  Every thread needs a context (stream, workspaces, pinned buffers); one whose thread has ended
    and that no frame is bound to is taken over by the next thread instead of piling up until exit."""
    import json
    _build()
    out = subprocess.check_output([os.path.join(HOST, "replay", "thread_churn"), "40"], text=True)
    d = json.loads(out.strip().splitlines()[-1])
    assert d["threads"] == 40 and d["failed"] == 0 and d["device_errors"] == 0, d
    assert d["contexts"] <= 3, d      # the main thread's and one or two that go round


@pytest.mark.gpu
@pytest.mark.parametrize("cell", ["0.25", "0.2"])
def test_node_replay_on_a_frame_of_a_million_cells(tmp_path, cell):
    """A 300 m frame of 0.25 m cells (1.44 M cells; the reference spends 11 GB of host memory on it): until round 6 the resident
    frame refused anything beyond ~650 000 cells because its table packer sized its LDS for the whole grid; it is the box of the
    BUILT cells that has to fit.  And of 0.2 m cells (2.25 M, beyond the 2^21 cells a resident frame can have): the drop-in keeps
    such a frame's cells on its own side of the C-ABI.  60 scans of the node sequence through the drop-in -- resident frames with
    and without clusters, host-kept frames -- and the oracle: every pose identical (tests/campaigns/soak_replay.py)."""
    import sys
    _build()
    env = dict(os.environ, SOAK_FRAME="300", SOAK_CELL=cell, SOAK_SCORE="exact")
    env.pop("NDTPSO_ABORT_ON_ERROR", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "campaigns", "soak_replay.py"), "60", "--oracle"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if "max |dpose|" in l]
    assert len(lines) == 4 and all("max |dpose| 0.000e+00" in l for l in lines), r.stdout[-1500:]
