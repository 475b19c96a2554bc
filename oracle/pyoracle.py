"""ctypes binding of the CPU oracle (oracle/ndtpso_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.  PARITY UNPINNED
(see oracle/ndtpso_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libndtpso_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("ndtpso_oracle.c", "ndtpso_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class PSOConfig(C.Structure):
    _fields_ = [("iterations", C.c_int), ("population", C.c_int), ("num_threads", C.c_int),
                ("w", C.c_double), ("c1", C.c_double), ("c2", C.c_double), ("w_damping", C.c_double)]

    @staticmethod
    def make(iterations=50, population=30, w=0.8, c1=2.0, c2=2.0, w_damping=1.0):
        return PSOConfig(iterations, population, -1, w, c1, c2, w_damping)


class Rand(C.Structure):
    _fields_ = [("table", C.POINTER(C.c_int32)), ("n", C.c_size_t), ("cursor", C.c_size_t)]


class PSOStats(C.Structure):
    _fields_ = [("cost_evals", C.c_uint64), ("pbest_updates", C.c_uint64),
                ("gbest_updates", C.c_uint64), ("rand_draws", C.c_uint64)]


class CellRow(C.Structure):
    _fields_ = [("index", C.c_int32), ("count", C.c_int32), ("built", C.c_int32), ("n_slot0", C.c_int32),
                ("window_id", C.c_int32), ("current_count", C.c_int32),
                ("mean", C.c_double * 2), ("icov", C.c_double * 4)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    dp = C.POINTER(C.c_double)
    fp = C.POINTER(C.c_float)
    L.orc_frame_create.restype = C.c_void_p
    L.orc_frame_create.argtypes = [dp, C.c_ushort, C.c_ushort, C.c_double, C.c_float]
    L.orc_frame_destroy.argtypes = [C.c_void_p]
    L.orc_frame_load_laser.argtypes = [C.c_void_p, fp, C.c_uint, C.c_float, C.c_float, C.c_float]
    L.orc_frame_add_point.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.orc_frame_update.argtypes = [C.c_void_p, dp, C.c_void_p]
    L.orc_frame_build.argtypes = [C.c_void_p]
    L.orc_frame_reset_cells.argtypes = [C.c_void_p]
    L.orc_frame_get_cell_index.restype = C.c_int
    L.orc_frame_get_cell_index.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.orc_frame_set_trans.argtypes = [C.c_void_p, dp]
    L.orc_frame_align.argtypes = [C.c_void_p, dp, C.c_void_p, C.POINTER(PSOConfig), C.POINTER(Rand), dp]
    L.orc_cost_function.restype = C.c_double
    L.orc_cost_function.argtypes = [dp, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.orc_pso_optimization.argtypes = [dp, C.c_void_p, C.c_void_p, dp, C.POINTER(PSOConfig), C.POINTER(Rand),
                                       dp, dp, C.POINTER(PSOStats)]
    L.orc_pso_optimization_omp.argtypes = [dp, C.c_void_p, C.c_void_p, dp, C.POINTER(PSOConfig), C.c_int, dp, dp]
    L.orc_frame_num_points.restype = C.c_uint
    L.orc_frame_num_points.argtypes = [C.c_void_p]
    L.orc_frame_get_points.restype = C.c_uint
    L.orc_frame_get_points.argtypes = [C.c_void_p, dp]
    L.orc_frame_get_points_all.restype = C.c_ulong
    L.orc_frame_get_points_all.argtypes = [C.c_void_p, dp, C.c_ulong]
    L.orc_frame_num_created.restype = C.c_uint
    L.orc_frame_num_created.argtypes = [C.c_void_p]
    L.orc_frame_export_cells.restype = C.c_uint
    L.orc_frame_export_cells.argtypes = [C.c_void_p, C.POINTER(CellRow), C.c_uint]
    L.orc_frame_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.orc_frame_enable_occupancy_grid.argtypes = [C.c_void_p, C.c_double]
    L.orc_frame_occupancy_grid.restype = C.POINTER(C.c_int8)
    L.orc_frame_occupancy_grid.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.orc_glibc_rand_fill.argtypes = [C.c_uint32, C.POINTER(C.c_int32), C.c_size_t]
    L.orc_eigen_eigenvalues_2x2.argtypes = [dp, C.c_int, dp]
    L.orc_set_eigen_variant.argtypes = [C.c_int]
    L.orc_get_eigen_variant.restype = C.c_int
    L.orc_covar_inverse_batch.argtypes = [dp, C.c_size_t, C.c_int, dp]
    L.orc_pso_rand_draws.restype = C.c_size_t
    L.orc_pso_rand_draws.argtypes = [C.POINTER(PSOConfig)]
    L.orc_align_pairs.restype = C.c_int
    L.orc_align_pairs.argtypes = [C.c_int, fp, fp, C.c_uint, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_ushort, C.c_ushort, C.c_double, dp, dp, C.POINTER(PSOConfig),
                                  C.POINTER(C.c_uint32), C.c_int, dp, dp]
    L.orc_libm_exp.argtypes = [dp, C.c_size_t, dp]
    L.orc_libm_exp.restype = None
    L.orc_libm_sincos.argtypes = [dp, C.c_size_t, dp, dp]
    L.orc_libm_sincos.restype = None
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))


def glibc_rand(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.int32)
    lib().orc_glibc_rand_fill(C.c_uint32(int(seed)), out.ctypes.data_as(C.POINTER(C.c_int32)), n)
    return out


EIGEN_337, EIGEN_NOSCALE, EIGEN_CLOSED_FORM = 0, 1, 2


def eigenvalues_2x2(m, variant=EIGEN_337):
    """EigenSolver<Matrix2d>(m).pseudoEigenvalueMatrix().diagonal() as the oracle restates it (ndtcell.cpp:96-97)."""
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(4))
    ev = np.empty(2)
    lib().orc_eigen_eigenvalues_2x2(_dp(a), int(variant), _dp(ev))
    return ev


def covar_inverse_batch(covars, variant=EIGEN_337):
    """ndtcell.cpp:93-111 on an (n, 4) array of covariance matrices -> (n, 8): large, small, det, degenerate, inv[4]."""
    a = np.ascontiguousarray(np.asarray(covars, dtype=np.float64).reshape(-1, 4))
    out = np.empty((a.shape[0], 8))
    lib().orc_covar_inverse_batch(_dp(a), a.shape[0], int(variant), _dp(out))
    return out


def set_eigen_variant(variant):
    lib().orc_set_eigen_variant(int(variant))


class Frame:
    """Thin wrapper of orc_frame (mirrors NDTFrame's methods on the path)."""

    def __init__(self, trans=(0.0, 0.0, 0.0), width=20, height=20, cell_side=1.0, laser_ignore_epsilon=0.1):
        self._h = lib().orc_frame_create(_dp(_vec3(trans)), width, height, float(cell_side),
                                         float(laser_ignore_epsilon))
        self.width, self.height, self.cell_side = width, height, float(cell_side)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_frame_destroy(self._h)
            self._h = None

    def load_laser(self, ranges, min_angle, angle_increment, max_range):
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        lib().orc_frame_load_laser(self._h, r.ctypes.data_as(C.POINTER(C.c_float)), r.size,
                                   float(min_angle), float(angle_increment), float(max_range))

    def add_point(self, x, y):
        lib().orc_frame_add_point(self._h, float(x), float(y))

    def update(self, trans, new_frame: "Frame"):
        lib().orc_frame_update(self._h, _dp(_vec3(trans)), new_frame._h)

    def build(self):
        lib().orc_frame_build(self._h)

    def reset_cells(self):
        lib().orc_frame_reset_cells(self._h)

    def enable_occupancy_grid(self, og_cell_size):
        lib().orc_frame_enable_occupancy_grid(self._h, float(og_cell_size))

    def occupancy_grid(self):
        """(og[height-major as the reference indexes it: og[x + height*y]], width, height, (min_x, max_x, min_y, max_y))"""
        w, h = C.c_uint32(), C.c_uint32()
        mm = (C.c_uint32 * 4)()
        ptr = lib().orc_frame_occupancy_grid(self._h, C.byref(w), C.byref(h), mm)
        og = np.ctypeslib.as_array(ptr, shape=(w.value * h.value,)).copy()
        return og, w.value, h.value, tuple(mm)

    def set_trans(self, trans):
        lib().orc_frame_set_trans(self._h, _dp(_vec3(trans)))

    def get_cell_index(self, x, y) -> int:
        return lib().orc_frame_get_cell_index(self._h, float(x), float(y))

    def dims(self):
        w, h = C.c_int32(), C.c_int32()
        lib().orc_frame_dims(self._h, C.byref(w), C.byref(h))
        return w.value, h.value

    def points(self) -> np.ndarray:
        n = lib().orc_frame_num_points(self._h)
        xy = np.empty((n, 2), dtype=np.float64)
        if n:
            lib().orc_frame_get_points(self._h, _dp(xy))
        return xy

    def points_all(self) -> np.ndarray:
        """every stored point of every window slot, in dumpMap's order"""
        n = lib().orc_frame_get_points_all(self._h, None, 0)
        xy = np.empty((max(n, 1), 2), dtype=np.float64)
        lib().orc_frame_get_points_all(self._h, _dp(xy), n)
        return xy[:n]

    def cells(self):
        n = lib().orc_frame_num_created(self._h)
        rows = (CellRow * max(n, 1))()
        m = lib().orc_frame_export_cells(self._h, rows, n)
        out = []
        for i in range(m):
            r = rows[i]
            out.append(dict(index=r.index, count=r.count, built=bool(r.built), n_slot0=r.n_slot0, slot=r.window_id,
                            current_count=r.current_count,
                            mean=np.array(r.mean[:]), icov=np.array(r.icov[:])))
        return out

    def cost(self, trans, new_frame: "Frame", want_cells=False):
        idx = None
        ptr = None
        if want_cells:
            idx = np.empty(lib().orc_frame_num_points(new_frame._h), dtype=np.int32)
            ptr = idx.ctypes.data_as(C.POINTER(C.c_int32))
        c = lib().orc_cost_function(_dp(_vec3(trans)), self._h, new_frame._h, ptr)
        return (c, idx) if want_cells else c

    def pso(self, guess, new_frame: "Frame", deviation, cfg: PSOConfig, seed=None, table=None):
        """pso_optimization with the srand(seed) stream (or an explicit rand() table)."""
        if table is None:
            table = glibc_rand(seed, lib().orc_pso_rand_draws(C.byref(cfg)))
        table = np.ascontiguousarray(table, dtype=np.int32)
        g = Rand(table.ctypes.data_as(C.POINTER(C.c_int32)), table.size, 0)
        pose = np.empty(3)
        cost = C.c_double()
        st = PSOStats()
        lib().orc_pso_optimization(_dp(_vec3(guess)), self._h, new_frame._h, _dp(_vec3(deviation)),
                                   C.byref(cfg), C.byref(g), _dp(pose), C.byref(cost), C.byref(st))
        stats = dict(cost_evals=st.cost_evals, pbest_updates=st.pbest_updates,
                     gbest_updates=st.gbest_updates, rand_draws=st.rand_draws)
        return pose, cost.value, stats

    def pso_omp(self, guess, new_frame: "Frame", deviation, cfg: PSOConfig, n_threads=0):
        """pso_optimization in the reference's parallel shape (OpenMP over particles, live rand(), racy gbest,
        core.cpp:72-109): irreproducible like the original -- for TIMING the CPU baseline only."""
        pose = np.empty(3)
        cost = C.c_double()
        lib().orc_pso_optimization_omp(_dp(_vec3(guess)), self._h, new_frame._h, _dp(_vec3(deviation)), C.byref(cfg),
                                       int(n_threads), _dp(pose), C.byref(cost))
        return pose, cost.value

    def align(self, guess, new_frame: "Frame", cfg: PSOConfig | None, seed=None, table=None):
        c = cfg if cfg is not None else PSOConfig.make()
        if table is None:
            table = glibc_rand(seed, lib().orc_pso_rand_draws(C.byref(c)))
        table = np.ascontiguousarray(table, dtype=np.int32)
        g = Rand(table.ctypes.data_as(C.POINTER(C.c_int32)), table.size, 0)
        pose = np.empty(3)
        lib().orc_frame_align(self._h, _dp(_vec3(guess)), new_frame._h,
                              C.byref(cfg) if cfg is not None else None, C.byref(g), _dp(pose))
        return pose


def align_pairs(ref_ranges, new_ranges, min_angle, angle_increment, max_range, eps, width, height, cell_side,
                guess, deviation, cfg: PSOConfig, seeds, n_threads=0):
    ref = np.ascontiguousarray(ref_ranges, dtype=np.float32)
    new = np.ascontiguousarray(new_ranges, dtype=np.float32)
    B, N = ref.shape
    guess = np.ascontiguousarray(np.broadcast_to(np.asarray(guess, dtype=np.float64), (B, 3)))
    deviation = np.ascontiguousarray(np.broadcast_to(np.asarray(deviation, dtype=np.float64), (B, 3)))
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    pose = np.empty((B, 3))
    cost = np.empty(B)
    fp = C.POINTER(C.c_float)
    used = lib().orc_align_pairs(B, ref.ctypes.data_as(fp), new.ctypes.data_as(fp), N, float(min_angle),
                                 float(angle_increment), float(max_range), float(eps), width, height,
                                 float(cell_side), _dp(guess), _dp(deviation), C.byref(cfg),
                                 seeds.ctypes.data_as(C.POINTER(C.c_uint32)), int(n_threads), _dp(pose), _dp(cost))
    return pose, cost, used


def libm_exp(x):
    """glibc's exp, element by element (numpy's own exp is a SIMD routine of its own)."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    out = np.empty_like(x)
    lib().orc_libm_exp(_dp(x), x.size, _dp(out))
    return out


def libm_sincos(x):
    """glibc's sincos, element by element: (sin, cos)."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    s, c = np.empty_like(x), np.empty_like(x)
    lib().orc_libm_sincos(_dp(x), x.size, _dp(s), _dp(c))
    return s, c
