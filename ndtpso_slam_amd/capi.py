"""ctypes binding of the C-ABI in include/ndtpso_hip.h (libndtpso_hip.so).

There is no CPU fallback: if the library is missing or no HIP device is usable every
call raises NdtpsoError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

SCORE_F32 = 0
SCORE_F64 = 1
SCORE_EXACT = 2   # fp32 score + fp64 arbitration of undecidable comparisons: SCORE_F64's results (include/ndtpso_hip.h)

OK, E_HIP, E_ARG, E_CAPACITY, E_STATE = 0, -1, -2, -3, -4


class NdtpsoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ndtpso error {code}: {msg}")
        self.code = code


class PSOConfig(C.Structure):
    """PSOConfig, include/ndtpso_slam/config.h:27-38 of the reference."""
    _fields_ = [("iterations", C.c_int32), ("population", C.c_int32), ("num_threads", C.c_int32),
                ("w", C.c_double), ("c1", C.c_double), ("c2", C.c_double), ("w_damping", C.c_double)]

    @staticmethod
    def make(iterations=50, population=30, w=0.8, c1=2.0, c2=2.0, w_damping=1.0):
        return PSOConfig(iterations, population, -1, w, c1, c2, w_damping)


class Grid(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("cell_side", C.c_double)]


class ScanGeom(C.Structure):
    _fields_ = [("n_beams", C.c_uint32), ("min_angle", C.c_float), ("angle_increment", C.c_float),
                ("max_range", C.c_float), ("laser_ignore_epsilon", C.c_float)]


class CellRow(C.Structure):
    _fields_ = [("index", C.c_int32), ("count", C.c_int32), ("built", C.c_int32), ("reserved", C.c_int32),
                ("mean", C.c_double * 2), ("icov", C.c_double * 4)]


class CellWindow(C.Structure):
    """ndtpso_cell_window: the NDTCell sliding-window state crossing the boundary (ndtcell.h:65-68)."""
    _fields_ = [("global_sum", C.c_double * 2), ("global_covar_sum", C.c_double * 4), ("slot_sum", C.c_double * 2),
                ("slot_covar", C.c_double * 4), ("mean", C.c_double * 2), ("icov", C.c_double * 4),
                ("global_count", C.c_int32), ("slot_count", C.c_int32), ("current_count", C.c_int32),
                ("built", C.c_int32)]


CELL_WINDOW_DTYPE = np.dtype([("global_sum", "<f8", (2,)), ("global_covar_sum", "<f8", (4,)), ("slot_sum", "<f8", (2,)),
                              ("slot_covar", "<f8", (4,)), ("mean", "<f8", (2,)), ("icov", "<f8", (4,)),
                              ("global_count", "<i4"), ("slot_count", "<i4"), ("current_count", "<i4"), ("built", "<i4")])


class AlignStats(C.Structure):
    _fields_ = [("n_points", C.c_uint32), ("n_built", C.c_uint32), ("cost_evals", C.c_uint32),
                ("rounds", C.c_uint32), ("gbest_updates", C.c_uint32), ("status", C.c_uint32),
                ("t_start", C.c_uint32), ("t_end", C.c_uint32)]


def _stats_dict(st: "AlignStats") -> dict:
    """The C struct as a dict, its `status` word split like STATS_DTYPE: flags (low half), SCORE_EXACT's count of
    arbitrated comparisons (high half)."""
    d = {k: getattr(st, k) for k, _ in AlignStats._fields_}
    d["arbitrated"] = d["status"] >> 16
    d["status"] &= 0xFFFF
    return d


class PairsPlan(C.Structure):
    _fields_ = [("lds_bytes", C.c_uint32), ("block_threads", C.c_uint32), ("workgroups_per_cu", C.c_uint32),
                ("table_form", C.c_uint32), ("swarm_in_hbm", C.c_uint32), ("window_w", C.c_uint32),
                ("window_h", C.c_uint32), ("table_bytes", C.c_uint32)]


class ShardInfo(C.Structure):
    """ndtpso_shard_info (include/ndtpso_hip.h)"""
    _fields_ = [("n_shards", C.c_int32), ("gather_kind", C.c_int32), ("rccl_version", C.c_int32), ("comm_ranks", C.c_int32),
                ("devices", C.c_int32 * 64)]


class MapInfo(C.Structure):
    _fields_ = [("n_created", C.c_uint32), ("n_built", C.c_uint32), ("status", C.c_uint32), ("n_words", C.c_uint32),
                ("x0", C.c_int32), ("x1", C.c_int32), ("y0", C.c_int32), ("y1", C.c_int32),
                ("og_min_x", C.c_uint32), ("og_max_x", C.c_uint32), ("og_min_y", C.c_uint32), ("og_max_y", C.c_uint32),
                ("pool_bump", C.c_int32), ("pool_free", C.c_int32), ("n_points", C.c_uint64)]


STATS_DTYPE = np.dtype([("n_points", "<u4"), ("n_built", "<u4"), ("cost_evals", "<u4"), ("rounds", "<u4"),
                        ("gbest_updates", "<u4"), ("status", "<u2"), ("arbitrated", "<u2"), ("t_start", "<u4"),
                        ("t_end", "<u4")])   # the C struct's `status` word: flags in its low half, SCORE_EXACT's count above
assert STATS_DTYPE.itemsize == C.sizeof(AlignStats)
assert CELL_WINDOW_DTYPE.itemsize == C.sizeof(CellWindow) == 160

EXPORTS = [
    "ndtpso_ctx_create", "ndtpso_ctx_destroy", "ndtpso_last_error", "ndtpso_set_stream", "ndtpso_synchronize",
    "ndtpso_set_pipeline_depth", "ndtpso_pipeline_flush",
    "ndtpso_rand_draws", "ndtpso_scan_to_points", "ndtpso_ref_from_points", "ndtpso_ref_from_scan",
    "ndtpso_ref_set_cells", "ndtpso_ref_get_cells", "ndtpso_points_to_cells", "ndtpso_scan_to_cells", "ndtpso_cells_build_windowed", "ndtpso_occupancy_values", "ndtpso_cost_batch", "ndtpso_align", "ndtpso_align_pairs",
    "ndtpso_align_pairs_dev", "ndtpso_align_pairs_footprint", "ndtpso_align_pairs_describe",
    "ndtpso_points_create", "ndtpso_points_destroy", "ndtpso_points_load_scan", "ndtpso_points_set", "ndtpso_points_get",
    "ndtpso_map_create", "ndtpso_map_destroy", "ndtpso_map_reset", "ndtpso_map_clear", "ndtpso_map_mark_unbuilt", "ndtpso_map_insert", "ndtpso_map_insert_host",
    "ndtpso_map_build", "ndtpso_map_speculate_build", "ndtpso_map_align", "ndtpso_map_cost", "ndtpso_map_get_info", "ndtpso_map_get_cells", "ndtpso_map_get_points",
    "ndtpso_map_get_occupancy",
    "ndtpso_shard_group_create", "ndtpso_shard_group_destroy", "ndtpso_shard_group_size", "ndtpso_shard_last_error",
    "ndtpso_shard_range", "ndtpso_align_pairs_sharded", "ndtpso_align_pairs_sharded_dev", "ndtpso_shard_last_timing",
    "ndtpso_shard_gathered", "ndtpso_shard_group_describe", "ndtpso_shard_verify_gather",
    "ndtpso_shard_last_gather_device_us",
    "ndtpso_selftest_exp", "ndtpso_device_math", "ndtpso_exact_check", "ndtpso_exact_check_report", "ndtpso_process_counters",
]

_lib = None


def library_path() -> str:
    return _build.LIB


class _MissingSymbol:
    def __init__(self, name):
        self._name = name

    def __call__(self, *a):
        raise NdtpsoError(E_HIP, "%s is not exported by the library NDTPSO_LIB names" % self._name)


class _TolerantCDLL(C.CDLL):
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if not name.startswith("ndtpso_"):
                raise
            stub = _MissingSymbol(name)
            setattr(self, name, stub)
            return stub


def load(build_if_missing: bool = True):
    """dlopen the HIP library (building it with hipcc first if it is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("NDTPSO_LIB") or _build.LIB   # NDTPSO_LIB: kernel-variant experiments
    if path == _build.LIB and build_if_missing and _build.needs_build():
        _build.build_hip()
    if not os.path.exists(path):
        raise NdtpsoError(E_HIP, f"{path} is missing: build it with `python -m ndtpso_slam_amd.build`")
    # (NDTPSO_LIB -- an experiment's or an older round's build, scripts/ab_libs.py -- may lack the newest entry points: they bind
    # to a stub that raises when called; the shipped library must export every one, tests/test_capi_exports.py)
    L = _TolerantCDLL(path) if os.environ.get("NDTPSO_LIB") else C.CDLL(path)
    vp, dp, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
    ip, up = C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    L.ndtpso_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ndtpso_ctx_destroy.argtypes = [vp]
    L.ndtpso_ctx_destroy.restype = None
    L.ndtpso_last_error.argtypes = [vp]
    L.ndtpso_last_error.restype = C.c_char_p
    L.ndtpso_set_stream.argtypes = [vp, vp]
    L.ndtpso_synchronize.argtypes = [vp]
    L.ndtpso_set_pipeline_depth.argtypes = [vp, C.c_int]
    L.ndtpso_pipeline_flush.argtypes = [vp, C.c_int]
    L.ndtpso_rand_draws.argtypes = [C.POINTER(PSOConfig)]
    L.ndtpso_rand_draws.restype = C.c_size_t
    L.ndtpso_scan_to_points.argtypes = [vp, fp, C.POINTER(ScanGeom), dp, dp, up]
    L.ndtpso_ref_from_points.argtypes = [vp, C.POINTER(Grid), dp, C.c_uint32]
    L.ndtpso_ref_from_scan.argtypes = [vp, C.POINTER(Grid), fp, C.POINTER(ScanGeom), dp]
    L.ndtpso_ref_set_cells.argtypes = [vp, C.POINTER(Grid), C.c_uint32, ip, dp, dp]
    L.ndtpso_ref_get_cells.argtypes = [vp, C.POINTER(CellRow), C.c_uint32, up]
    L.ndtpso_points_to_cells.argtypes = [vp, C.POINTER(Grid), dp, C.c_uint32, dp, dp, ip]
    L.ndtpso_scan_to_cells.argtypes = [vp, fp, C.POINTER(ScanGeom), dp, C.POINTER(Grid), dp, ip, up]
    L.ndtpso_cells_build_windowed.argtypes = [vp, C.c_uint32, vp, up, dp]
    L.ndtpso_occupancy_values.argtypes = [vp, C.POINTER(Grid), C.c_double, C.c_uint32, ip, dp, dp, C.POINTER(C.c_int8)]
    L.ndtpso_cost_batch.argtypes = [vp, dp, C.c_uint32, dp, C.c_uint32, C.c_int, dp, ip]
    L.ndtpso_align.argtypes = [vp, dp, C.c_uint32, dp, dp, C.POINTER(PSOConfig), C.c_uint32, ip, C.c_int, dp, dp,
                               C.POINTER(AlignStats)]
    pairs_args = [vp, C.c_uint32, vp, vp, C.POINTER(ScanGeom), C.POINTER(Grid), vp, vp, C.POINTER(PSOConfig),
                  vp, vp, C.c_int, vp, vp, vp]
    L.ndtpso_align_pairs.argtypes = pairs_args
    L.ndtpso_align_pairs_dev.argtypes = pairs_args
    L.ndtpso_align_pairs_describe.argtypes = [C.POINTER(ScanGeom), C.POINTER(Grid), C.POINTER(PSOConfig), C.c_int,
                                              C.c_uint32, C.POINTER(PairsPlan)]
    L.ndtpso_align_pairs_footprint.argtypes = [C.POINTER(ScanGeom), C.POINTER(Grid), C.POINTER(PSOConfig), up, up]
    L.ndtpso_selftest_exp.argtypes = [vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64), dp]
    L.ndtpso_device_math.argtypes = [vp, C.c_int, dp, C.c_uint32, dp, dp]
    L.ndtpso_exact_check.argtypes = [vp, C.POINTER(C.c_int), up, up, dp]
    L.ndtpso_exact_check_report.argtypes = [vp, C.c_char_p, C.c_uint32]
    L.ndtpso_process_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    L.ndtpso_points_create.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.ndtpso_points_destroy.argtypes = [vp]
    L.ndtpso_points_destroy.restype = None
    L.ndtpso_points_load_scan.argtypes = [vp, fp, C.POINTER(ScanGeom), dp, C.POINTER(Grid), C.c_int]
    L.ndtpso_points_set.argtypes = [vp, dp, C.c_uint32]
    L.ndtpso_points_get.argtypes = [vp, dp, C.c_uint32, up]
    L.ndtpso_map_create.argtypes = [vp, C.POINTER(Grid), C.c_double, C.c_uint64, C.POINTER(vp)]
    L.ndtpso_map_destroy.argtypes = [vp]
    L.ndtpso_map_destroy.restype = None
    L.ndtpso_map_reset.argtypes = [vp]
    L.ndtpso_map_clear.argtypes = [vp]
    L.ndtpso_map_mark_unbuilt.argtypes = [vp]
    L.ndtpso_map_insert.argtypes = [vp, vp, dp]
    L.ndtpso_map_insert_host.argtypes = [vp, dp, C.c_uint32, dp]
    L.ndtpso_map_build.argtypes = [vp]
    L.ndtpso_map_speculate_build.argtypes = [vp]
    L.ndtpso_map_align.argtypes = [vp, vp, dp, dp, C.POINTER(PSOConfig), C.c_uint32, ip, C.c_int, dp, dp,
                                   C.POINTER(AlignStats)]
    L.ndtpso_map_cost.argtypes = [vp, vp, dp, C.c_uint32, C.c_int, dp]
    L.ndtpso_map_get_info.argtypes = [vp, C.POINTER(MapInfo)]
    L.ndtpso_map_get_cells.argtypes = [vp, C.POINTER(CellRow), C.c_uint32, up]
    L.ndtpso_map_get_points.argtypes = [vp, C.c_int, dp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.ndtpso_map_get_occupancy.argtypes = [vp, C.POINTER(C.c_int8), C.c_uint64, up, up, up]
    L.ndtpso_shard_group_create.argtypes = [ip, C.c_int, C.POINTER(vp)]
    L.ndtpso_shard_group_destroy.argtypes = [vp]
    L.ndtpso_shard_group_destroy.restype = None
    L.ndtpso_shard_group_size.argtypes = [vp]
    L.ndtpso_shard_last_error.argtypes = [vp]
    L.ndtpso_shard_last_error.restype = C.c_char_p
    L.ndtpso_shard_range.argtypes = [C.c_uint32, C.c_int, C.c_int, up, up]
    L.ndtpso_shard_range.restype = None
    L.ndtpso_align_pairs_sharded.argtypes = [vp, C.c_uint32, fp, fp, C.POINTER(ScanGeom), C.POINTER(Grid), dp, dp,
                                             C.POINTER(PSOConfig), up, ip, C.c_int, dp, dp, vp]
    pp = C.POINTER(C.c_void_p)
    L.ndtpso_align_pairs_sharded_dev.argtypes = [vp, C.c_uint32, pp, pp, C.POINTER(ScanGeom), C.POINTER(Grid), pp, pp,
                                                 C.POINTER(PSOConfig), pp, pp, C.c_int, dp, dp, vp]
    L.ndtpso_shard_last_timing.argtypes = [vp, dp, dp]
    L.ndtpso_shard_gathered.argtypes = [vp, C.c_int]
    L.ndtpso_shard_group_describe.argtypes = [vp, C.POINTER(ShardInfo)]
    L.ndtpso_shard_verify_gather.argtypes = [vp, C.POINTER(C.c_int)]
    L.ndtpso_shard_last_gather_device_us.argtypes = [vp]
    L.ndtpso_shard_last_gather_device_us.restype = C.c_double
    L.ndtpso_shard_gathered.restype = C.c_void_p
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("ndtpso_ctx_destroy", "ndtpso_last_error", "ndtpso_rand_draws", "ndtpso_points_destroy",
                        "ndtpso_map_destroy", "ndtpso_shard_group_destroy", "ndtpso_shard_last_error", "ndtpso_shard_range",
                        "ndtpso_shard_gathered", "ndtpso_shard_last_gather_device_us"):
            fn.restype = C.c_int
    _lib = L
    return L


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Context:
    """One HIP device context (one per host thread / stream)."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.ndtpso_ctx_create(int(device), C.byref(h))
        if rc != OK:
            raise NdtpsoError(rc, "no usable HIP device (ndtpso_ctx_create failed); there is no CPU fallback")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndtpso_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise NdtpsoError(rc, self._lib.ndtpso_last_error(self._h).decode())

    def set_stream(self, stream_handle: int | None):
        self._chk(self._lib.ndtpso_set_stream(self._h, C.c_void_p(stream_handle or 0)))

    def synchronize(self):
        self._chk(self._lib.ndtpso_synchronize(self._h))

    def set_pipeline_depth(self, depth: int):
        """1: one batch at a time on the context's stream; 2: consecutive align_pairs_dev calls overlap on the device
        (ndtpso_set_pipeline_depth) -- outputs are ordered on the context's stream by pipeline_flush / synchronize."""
        self._chk(self._lib.ndtpso_set_pipeline_depth(self._h, int(depth)))

    def pipeline_flush(self, keep_newest: int = 0):
        self._chk(self._lib.ndtpso_pipeline_flush(self._h, int(keep_newest)))

    # ---- probes of the device arithmetic (tests/test_gpu_exp.py) ----
    def selftest_exp(self, exp2_lo: int, exp2_hi: int, per_binade: int, seed: int = 1, positive: bool = False):
        """(checked, mismatched, one differing argument or NaN): exp_neg_half against the library's exp on the device."""
        n, bad, x = C.c_uint64(), C.c_uint64(), C.c_double()
        self._chk(self._lib.ndtpso_selftest_exp(self._h, exp2_lo, exp2_hi, per_binade, seed, 1 if positive else 0,
                                                C.byref(n), C.byref(bad), C.byref(x)))
        return n.value, bad.value, x.value

    def device_math(self, kind: str, x):
        """'exp' / 'exp_neg_half' -> values; 'sincos' -> (sin, cos): the device's own arithmetic on x."""
        x = _f64(x).ravel()
        k = {"exp": 0, "exp_neg_half": 1, "sincos": 2}[kind]
        o0, o1 = np.empty_like(x), np.empty_like(x)
        self._chk(self._lib.ndtpso_device_math(self._h, k, _p(x, C.c_double), x.size, _p(o0, C.c_double), _p(o1, C.c_double)))
        return (o0, o1) if k == 2 else o0

    def exact_check(self):
        """The start-up known-answer check of the exact mode (run now if it has not run): dict(state 1 passed / 2 refused,
        arbitrated_batch, arbitrated_single, ms)."""
        st, a, b, ms = C.c_int(), C.c_uint32(), C.c_uint32(), C.c_double()
        self._chk(self._lib.ndtpso_exact_check(self._h, C.byref(st), C.byref(a), C.byref(b), C.byref(ms)))
        return dict(state=st.value, arbitrated_batch=a.value, arbitrated_single=b.value, ms=ms.value)

    def exact_check_report(self) -> dict:
        """ndtpso_exact_check_report: the check's verdict per kernel family (dict: state, ms, families[...])."""
        import json
        buf = C.create_string_buffer(16384)
        n = self._lib.ndtpso_exact_check_report(self._h, buf, len(buf))
        if n < 0:
            self._chk(n)
        return json.loads(buf.value.decode())

    # ---- K3 ----
    def scan_to_points(self, ranges, geom: ScanGeom, trans=(0.0, 0.0, 0.0)) -> np.ndarray:
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        assert r.size == geom.n_beams
        xy = np.empty((r.size, 2))
        n = C.c_uint32()
        self._chk(self._lib.ndtpso_scan_to_points(self._h, _p(r, C.c_float), C.byref(geom),
                                                  _p(_f64(trans, 3), C.c_double), _p(xy, C.c_double), C.byref(n)))
        return xy[:n.value].copy()

    def ref_from_points(self, grid: Grid, xy):
        xy = _f64(xy).reshape(-1, 2)
        self._chk(self._lib.ndtpso_ref_from_points(self._h, C.byref(grid), _p(xy, C.c_double), xy.shape[0]))

    def ref_from_scan(self, grid: Grid, ranges, geom: ScanGeom, trans=(0.0, 0.0, 0.0)):
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        self._chk(self._lib.ndtpso_ref_from_scan(self._h, C.byref(grid), _p(r, C.c_float), C.byref(geom),
                                                 _p(_f64(trans, 3), C.c_double)))

    def ref_set_cells(self, grid: Grid, index, mean, icov):
        index = np.ascontiguousarray(index, dtype=np.int32)
        mean = _f64(mean).reshape(-1, 2)
        icov = _f64(icov).reshape(-1, 4)
        self._chk(self._lib.ndtpso_ref_set_cells(self._h, C.byref(grid), index.size, _p(index, C.c_int32),
                                                 _p(mean, C.c_double), _p(icov, C.c_double)))

    def ref_get_cells(self):
        n = C.c_uint32()
        self._chk(self._lib.ndtpso_ref_get_cells(self._h, None, 0, C.byref(n)))
        rows = (CellRow * max(n.value, 1))()
        self._chk(self._lib.ndtpso_ref_get_cells(self._h, rows, n.value, C.byref(n)))
        return [dict(index=r.index, count=r.count, built=bool(r.built), mean=np.array(r.mean[:]),
                     icov=np.array(r.icov[:])) for r in rows[:n.value]]

    def points_to_cells(self, grid: Grid, xy, trans=None):
        """transform_point + getCellIndex for a point list (NDTFrame::update / addPoint)."""
        xy = _f64(xy).reshape(-1, 2)
        out = np.empty_like(xy)
        idx = np.empty(xy.shape[0], dtype=np.int32)
        t = _f64(trans, 3) if trans is not None else None
        self._chk(self._lib.ndtpso_points_to_cells(self._h, C.byref(grid), _p(xy, C.c_double), xy.shape[0],
                                                   _p(t, C.c_double) if t is not None else None,
                                                   _p(out, C.c_double), _p(idx, C.c_int32)))
        return out, idx

    def scan_to_cells(self, ranges, geom: ScanGeom, grid: Grid, trans=(0.0, 0.0, 0.0)):
        """NDTFrame::loadLaser in one launch: surviving points (beam order) and the cell each falls in."""
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        xy = np.empty((r.size, 2))
        idx = np.empty(r.size, dtype=np.int32)
        n = C.c_uint32()
        self._chk(self._lib.ndtpso_scan_to_cells(self._h, _p(r, C.c_float), C.byref(geom), _p(_f64(trans, 3), C.c_double),
                                                 C.byref(grid), _p(xy, C.c_double), _p(idx, C.c_int32), C.byref(n)))
        return xy[:n.value].copy(), idx[:n.value].copy()

    def cells_build_windowed(self, cells: np.ndarray, pts_offset, pts_xy):
        """NDTCell::build with window state for many cells; `cells` (CELL_WINDOW_DTYPE) is updated in place."""
        assert cells.dtype == CELL_WINDOW_DTYPE and cells.flags.c_contiguous
        off = np.ascontiguousarray(pts_offset, dtype=np.uint32)
        xy = _f64(pts_xy).reshape(-1, 2)
        assert off.size == cells.size + 1 and off[-1] == xy.shape[0]
        self._chk(self._lib.ndtpso_cells_build_windowed(self._h, cells.size, cells.ctypes.data_as(C.c_void_p),
                                                        _p(off, C.c_uint32), _p(xy, C.c_double)))
        return cells

    def occupancy_values(self, grid: Grid, og_cell_size, index, mean, icov):
        """int8(100 p) at the k x k sub-cell centres of each built cell ([n_cells, k, k]; -1 where p == 0)."""
        index = np.ascontiguousarray(index, dtype=np.int32)
        mean = _f64(mean).reshape(-1, 2)
        icov = _f64(icov).reshape(-1, 4)
        k = int(np.floor(grid.cell_side / og_cell_size))
        out = np.empty((index.size, k, k), dtype=np.int8)
        self._chk(self._lib.ndtpso_occupancy_values(self._h, C.byref(grid), float(og_cell_size), index.size,
                                                    _p(index, C.c_int32), _p(mean, C.c_double), _p(icov, C.c_double),
                                                    out.ctypes.data_as(C.POINTER(C.c_int8))))
        return out

    # ---- K1 ----
    def cost_batch(self, xy, poses, mode=SCORE_F32, want_cells=False):
        xy = _f64(xy).reshape(-1, 2)
        poses = _f64(poses).reshape(-1, 3)
        costs = np.empty(poses.shape[0])
        idx = np.empty((poses.shape[0], xy.shape[0]), dtype=np.int32) if want_cells else None
        self._chk(self._lib.ndtpso_cost_batch(self._h, _p(xy, C.c_double), xy.shape[0], _p(poses, C.c_double),
                                              poses.shape[0], mode, _p(costs, C.c_double),
                                              _p(idx, C.c_int32) if want_cells else None))
        return (costs, idx) if want_cells else costs

    # ---- K2 ----
    def align(self, xy, guess, deviation, cfg: PSOConfig, seed=1, rand_table=None, mode=SCORE_F32):
        xy = _f64(xy).reshape(-1, 2)
        pose = np.empty(3)
        cost = C.c_double()
        st = AlignStats()
        tab = None
        if rand_table is not None:
            tab = np.ascontiguousarray(rand_table, dtype=np.int32)
            assert tab.size >= self._lib.ndtpso_rand_draws(C.byref(cfg))
        self._chk(self._lib.ndtpso_align(self._h, _p(xy, C.c_double), xy.shape[0], _p(_f64(guess, 3), C.c_double),
                                         _p(_f64(deviation, 3), C.c_double), C.byref(cfg), C.c_uint32(int(seed)),
                                         _p(tab, C.c_int32) if tab is not None else None, mode,
                                         _p(pose, C.c_double), C.byref(cost), C.byref(st)))
        return pose, cost.value, _stats_dict(st)

    # ---- fused pairs ----
    def align_pairs(self, ref_ranges, new_ranges, geom: ScanGeom, grid: Grid, guess, deviation, cfg: PSOConfig,
                    seeds=None, rand_tables=None, mode=SCORE_F32):
        ref = np.ascontiguousarray(ref_ranges, dtype=np.float32)
        new = np.ascontiguousarray(new_ranges, dtype=np.float32)
        B = ref.shape[0]
        assert ref.shape == new.shape == (B, geom.n_beams)
        guess = np.ascontiguousarray(np.broadcast_to(_f64(guess), (B, 3)))
        deviation = np.ascontiguousarray(np.broadcast_to(_f64(deviation), (B, 3)))
        pose = np.empty((B, 3))
        cost = np.empty(B)
        stats = np.zeros(B, dtype=STATS_DTYPE)
        sd = np.ascontiguousarray(seeds, dtype=np.uint32) if seeds is not None else None
        tb = np.ascontiguousarray(rand_tables, dtype=np.int32) if rand_tables is not None else None
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
        self._chk(self._lib.ndtpso_align_pairs(self._h, B, vp(ref), vp(new), C.byref(geom), C.byref(grid), vp(guess),
                                               vp(deviation), C.byref(cfg), vp(sd), vp(tb), mode, vp(pose), vp(cost),
                                               vp(stats)))
        return pose, cost, stats

    def align_pairs_dev(self, n_pairs, d_ref, d_new, geom, grid, d_guess, d_dev, cfg, d_seeds, d_tables, mode,
                        d_pose, d_cost, d_stats):
        """Device pointers (ints); asynchronous on the context stream."""
        vp = lambda a: C.c_void_p(int(a)) if a else None  # noqa: E731
        self._chk(self._lib.ndtpso_align_pairs_dev(self._h, int(n_pairs), vp(d_ref), vp(d_new), C.byref(geom),
                                                   C.byref(grid), vp(d_guess), vp(d_dev), C.byref(cfg), vp(d_seeds),
                                                   vp(d_tables), mode, vp(d_pose), vp(d_cost), vp(d_stats)))


class ShardGroup:
    """ndtpso_shard_group: one batch of scan pairs on several devices from one process -- contiguous index ranges, one
    context and stream per device, ONE ncclAllGather of the poses (include/ndtpso_hip.h).  Raises when RCCL or a device
    is missing: nothing is computed anywhere else."""

    def __init__(self, devices):
        self._lib = load()
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        h = C.c_void_p()
        rc = self._lib.ndtpso_shard_group_create(_p(devs, C.c_int32), int(devs.size), C.byref(h))
        if rc != OK:
            raise NdtpsoError(rc, "ndtpso_shard_group_create failed (devices %s): a device or RCCL is missing" % list(devs))
        self._h = h

    def size(self) -> int:
        return int(self._lib.ndtpso_shard_group_size(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndtpso_shard_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def align_pairs(self, ref_ranges, new_ranges, geom: ScanGeom, grid: Grid, guess, deviation, cfg: PSOConfig,
                    seeds=None, rand_tables=None, mode=SCORE_EXACT):
        ref = np.ascontiguousarray(ref_ranges, dtype=np.float32)
        new = np.ascontiguousarray(new_ranges, dtype=np.float32)
        B = ref.shape[0]
        assert ref.shape == new.shape == (B, geom.n_beams)
        guess = np.ascontiguousarray(np.broadcast_to(_f64(guess), (B, 3)))
        deviation = np.ascontiguousarray(np.broadcast_to(_f64(deviation), (B, 3)))
        pose, cost, stats = np.empty((B, 3)), np.empty(B), np.zeros(B, dtype=STATS_DTYPE)
        sd = np.ascontiguousarray(seeds, dtype=np.uint32) if seeds is not None else None
        tb = np.ascontiguousarray(rand_tables, dtype=np.int32) if rand_tables is not None else None
        rc = self._lib.ndtpso_align_pairs_sharded(
            self._h, B, _p(ref, C.c_float), _p(new, C.c_float), C.byref(geom), C.byref(grid), _p(guess, C.c_double),
            _p(deviation, C.c_double), C.byref(cfg), _p(sd, C.c_uint32) if sd is not None else None,
            _p(tb, C.c_int32) if tb is not None else None, mode, _p(pose, C.c_double), _p(cost, C.c_double),
            stats.ctypes.data_as(C.c_void_p))
        if rc != OK:
            raise NdtpsoError(rc, self._lib.ndtpso_shard_last_error(self._h).decode())
        return pose, cost, stats

    def align_pairs_dev(self, n_pairs, d_ref, d_new, geom: ScanGeom, grid: Grid, d_guess, d_dev, cfg: PSOConfig, d_seeds=None,
                        d_tables=None, mode=SCORE_EXACT, fetch=True):
        """ndtpso_align_pairs_sharded_dev: every argument a list of G device pointers (ints), entry d = shard d's slice
        resident on device d and complete.  fetch=False leaves the results on the devices (gathered())."""
        G = self.size()

        def arr(ptrs):
            if ptrs is None:
                return None
            assert len(ptrs) == G
            return (C.c_void_p * G)(*[C.c_void_p(int(x)) if x else None for x in ptrs])

        B = int(n_pairs)
        pose, cost, stats = np.empty((B, 3)), np.empty(B), np.zeros(B, dtype=STATS_DTYPE)
        rc = self._lib.ndtpso_align_pairs_sharded_dev(
            self._h, B, arr(d_ref), arr(d_new), C.byref(geom), C.byref(grid), arr(d_guess), arr(d_dev), C.byref(cfg),
            arr(d_seeds), arr(d_tables), mode, _p(pose, C.c_double) if fetch else None, _p(cost, C.c_double) if fetch else None,
            stats.ctypes.data_as(C.c_void_p))
        if rc != OK:
            raise NdtpsoError(rc, self._lib.ndtpso_shard_last_error(self._h).decode())
        return (pose, cost, stats) if fetch else (None, None, stats)

    def describe(self) -> dict:
        """ndtpso_shard_group_describe: shards, how the poses are gathered, RCCL's version, ranks of the communicator."""
        info = ShardInfo()
        rc = self._lib.ndtpso_shard_group_describe(self._h, C.byref(info))
        if rc != OK:
            raise NdtpsoError(rc, "ndtpso_shard_group_describe")
        v = int(info.rccl_version)
        return {"n_shards": int(info.n_shards), "gather": "host-staged (test)" if info.gather_kind else "ncclAllGather (RCCL)",
                "rccl_version_code": v, "rccl_version": "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v else None,
                "comm_ranks": int(info.comm_ranks), "devices": [int(info.devices[i]) for i in range(int(info.n_shards))]}

    def verify_gather(self) -> int:
        """ndtpso_shard_verify_gather: shards whose poses every shard's gathered copy holds bit for bit (after a call)."""
        n = C.c_int(0)
        rc = self._lib.ndtpso_shard_verify_gather(self._h, C.byref(n))
        if rc != OK:
            raise NdtpsoError(rc, self._lib.ndtpso_shard_last_error(self._h).decode())
        return int(n.value)

    def last_gather_device_us(self) -> float:
        return float(self._lib.ndtpso_shard_last_gather_device_us(self._h))

    def gathered(self, index: int) -> int:
        """Device pointer of device `index`'s copy of the gathered batch (G blocks of [pose M x 3 | cost M])."""
        return int(self._lib.ndtpso_shard_gathered(self._h, int(index)) or 0)

    def last_timing(self):
        """Host-side microseconds of the last call: ({device: (start, uploads, launches)}, (enqueued, collective, total))."""
        G = self.size()
        per, call = np.zeros((G, 3)), np.zeros(3)
        self._lib.ndtpso_shard_last_timing(self._h, _p(per, C.c_double), _p(call, C.c_double))
        return per, call


def process_counters() -> dict:
    """ndtpso_process_counters: cluster timeouts, waits that slept, the CPU budget, threads waiting now."""
    v = (C.c_uint64 * 6)()
    rc = load().ndtpso_process_counters(v, 6)
    if rc != OK:
        raise NdtpsoError(rc, "ndtpso_process_counters")
    return {"cluster_timeouts": int(v[0]), "polite_waits": int(v[1]), "cpu_budget": int(v[2]), "waiting_now": int(v[3]),
            "alignments_kept_on_one_workgroup": int(v[4]), "batches_redone_on_clusters": int(v[5])}


def shard_range(n_pairs: int, rank: int, n_devices: int):
    a, b = C.c_uint32(), C.c_uint32()
    load().ndtpso_shard_range(int(n_pairs), int(rank), int(n_devices), C.byref(a), C.byref(b))
    return a.value, b.value


def align_pairs_footprint(geom: ScanGeom, grid: Grid, cfg: PSOConfig):
    L = load()
    lds, thr = C.c_uint32(), C.c_uint32()
    rc = L.ndtpso_align_pairs_footprint(C.byref(geom), C.byref(grid), C.byref(cfg), C.byref(lds), C.byref(thr))
    return rc, lds.value, thr.value


def align_pairs_describe(geom: ScanGeom, grid: Grid, cfg: PSOConfig, mode=SCORE_F32, n_pairs=512):
    """(rc, dict) -- how the fused kernel would be launched for this configuration (no GPU needed)."""
    L = load()
    pl = PairsPlan()
    rc = L.ndtpso_align_pairs_describe(C.byref(geom), C.byref(grid), C.byref(cfg), mode, n_pairs, C.byref(pl))
    return rc, {k: getattr(pl, k) for k, _ in PairsPlan._fields_}


class ResidentScan:
    """ndtpso_points: a loaded scan that stays on the device (the node's one-cell per-scan frame)."""

    def __init__(self, ctx: Context, capacity: int):
        self._ctx = ctx
        self._lib = ctx._lib
        h = C.c_void_p()
        ctx._chk(self._lib.ndtpso_points_create(ctx._h, int(capacity), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self._ctx, "_h", None):   # a context that is already gone took the device with it
                self._lib.ndtpso_points_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_scan(self, ranges, geom: ScanGeom, trans=None, clip: Grid | None = None, append=False):
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        assert r.size == geom.n_beams
        t = _p(_f64(trans, 3), C.c_double) if trans is not None else None
        self._ctx._chk(self._lib.ndtpso_points_load_scan(self._h, _p(r, C.c_float), C.byref(geom), t,
                                                         C.byref(clip) if clip is not None else None, int(bool(append))))

    def set(self, xy):
        xy = _f64(xy).reshape(-1, 2)
        self._ctx._chk(self._lib.ndtpso_points_set(self._h, _p(xy, C.c_double), xy.shape[0]))

    def get(self) -> np.ndarray:
        n = C.c_uint32()
        self._ctx._chk(self._lib.ndtpso_points_get(self._h, None, 0, C.byref(n)))
        xy = np.empty((max(n.value, 1), 2))
        self._ctx._chk(self._lib.ndtpso_points_get(self._h, _p(xy, C.c_double), n.value, C.byref(n)))
        return xy[:n.value].copy()


class ResidentMap:
    """ndtpso_map: the reference frame with its sliding-window cells, resident in HBM."""

    def __init__(self, ctx: Context, grid: Grid, og_cell_size=0.0, pool_bytes=0):
        self._ctx = ctx
        self._lib = ctx._lib
        self.grid = grid
        h = C.c_void_p()
        ctx._chk(self._lib.ndtpso_map_create(ctx._h, C.byref(grid), float(og_cell_size), int(pool_bytes), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self._ctx, "_h", None):
                self._lib.ndtpso_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        """NDTFrame::resetCells"""
        self._ctx._chk(self._lib.ndtpso_map_reset(self._h))

    def clear(self):
        self._ctx._chk(self._lib.ndtpso_map_clear(self._h))

    def mark_unbuilt(self):
        """the unconditional `built = false` of NDTFrame::update / loadLaser"""
        self._ctx._chk(self._lib.ndtpso_map_mark_unbuilt(self._h))

    def insert(self, scan: ResidentScan, pose=None):
        """pose given: NDTFrame::update(pose, scan frame); None: addPoint for every point of the scan"""
        if pose is not None:
            self.mark_unbuilt()
        self._ctx._chk(self._lib.ndtpso_map_insert(self._h, scan._h,
                                                   _p(_f64(pose, 3), C.c_double) if pose is not None else None))

    def insert_host(self, xy, pose=None):
        """pose given: NDTFrame::update(pose, a frame holding xy); None: NDTFrame::addPoint for every point"""
        xy = _f64(xy).reshape(-1, 2)
        if pose is not None:
            self.mark_unbuilt()
        self._ctx._chk(self._lib.ndtpso_map_insert_host(self._h, _p(xy, C.c_double), xy.shape[0],
                                                        _p(_f64(pose, 3), C.c_double) if pose is not None else None))

    def build(self):
        self._ctx._chk(self._lib.ndtpso_map_build(self._h))

    def speculate_build(self):
        self._ctx._chk(self._lib.ndtpso_map_speculate_build(self._h))

    def align(self, scan: ResidentScan, guess, deviation, cfg: PSOConfig, seed=1, rand_table=None, mode=SCORE_F32):
        pose = np.empty(3)
        cost = C.c_double()
        st = AlignStats()
        tab = None
        if rand_table is not None:
            tab = np.ascontiguousarray(rand_table, dtype=np.int32)
            assert tab.size >= self._lib.ndtpso_rand_draws(C.byref(cfg))
        self._ctx._chk(self._lib.ndtpso_map_align(self._h, scan._h, _p(_f64(guess, 3), C.c_double),
                                                  _p(_f64(deviation, 3), C.c_double), C.byref(cfg),
                                                  C.c_uint32(int(seed)), _p(tab, C.c_int32) if tab is not None else None,
                                                  mode, _p(pose, C.c_double), C.byref(cost), C.byref(st)))
        return pose, cost.value, _stats_dict(st)

    def cost(self, scan: ResidentScan, poses, mode=SCORE_F32) -> np.ndarray:
        poses = _f64(poses).reshape(-1, 3)
        costs = np.empty(poses.shape[0])
        self._ctx._chk(self._lib.ndtpso_map_cost(self._h, scan._h, _p(poses, C.c_double), poses.shape[0], mode,
                                                 _p(costs, C.c_double)))
        return costs

    def info(self) -> dict:
        i = MapInfo()
        self._ctx._chk(self._lib.ndtpso_map_get_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in MapInfo._fields_}

    def cells(self):
        n = C.c_uint32()
        self._ctx._chk(self._lib.ndtpso_map_get_cells(self._h, None, 0, C.byref(n)))
        rows = (CellRow * max(n.value, 1))()
        self._ctx._chk(self._lib.ndtpso_map_get_cells(self._h, rows, n.value, C.byref(n)))
        return [dict(index=r.index, count=r.count, built=bool(r.built), slot=r.reserved, mean=np.array(r.mean[:]),
                     icov=np.array(r.icov[:])) for r in rows[:n.value]]

    def points(self, slot0_only=False) -> np.ndarray:
        n = C.c_uint64()
        self._ctx._chk(self._lib.ndtpso_map_get_points(self._h, int(slot0_only), None, 0, C.byref(n)))
        xy = np.empty((max(n.value, 1), 2))
        self._ctx._chk(self._lib.ndtpso_map_get_points(self._h, int(slot0_only), _p(xy, C.c_double), n.value, C.byref(n)))
        return xy[:n.value].copy()

    def occupancy(self):
        w, h = C.c_uint32(), C.c_uint32()
        ext = (C.c_uint32 * 4)()
        self._ctx._chk(self._lib.ndtpso_map_get_occupancy(self._h, None, 0, C.byref(w), C.byref(h), ext))
        og = np.zeros(max(w.value * h.value, 1), dtype=np.int8)
        self._ctx._chk(self._lib.ndtpso_map_get_occupancy(self._h, og.ctypes.data_as(C.POINTER(C.c_int8)), og.size,
                                                          C.byref(w), C.byref(h), ext))
        return og[:w.value * h.value], w.value, h.value, tuple(ext)
