"""Second, independent restatement of the reference arithmetic in plain numpy / Python loops.

Written separately from oracle/ndtpso_oracle.c (different data structures: dictionaries of point lists),
used only to cross-check the C oracle on small cases.  Citations are reference file:line.
"""
import ctypes
import ctypes.util
import math

import numpy as np

RAND_MAX = 2147483647

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
_libm.sincos.restype = None


def cos_sin(theta):
    """(cos, sin) of one angle from ONE glibc sincos() call: what GCC, the reference's compiler, makes of cos(x) and
    sin(x) of the same argument (transform_point core.h:28-31, laser_to_point core.h:45-47).  glibc's sincos differs
    from its cos / sin in the last bit for about 0.15 % of arguments, so math.cos / math.sin are not a faithful
    restatement."""
    s, c = ctypes.c_double(), ctypes.c_double()
    _libm.sincos(float(theta), ctypes.byref(s), ctypes.byref(c))
    return c.value, s.value


def uniform_pm1(raw):
    """Eigen DenseBase::Random() for double: x + (y-x)*double(rand())/double(RAND_MAX), x=-1, y=1."""
    return -1.0 + (2.0 * float(raw)) / float(RAND_MAX)


def laser_points(ranges, amin, ainc, rmax, eps=0.1, trans=(0.0, 0.0, 0.0)):
    """NDTFrame::loadLaser point generation, ndtframe.cpp:144-185 + core.h:40-47."""
    out = []
    amin, ainc, rmax, eps = np.float32(amin), np.float32(ainc), np.float32(rmax), np.float32(eps)
    do_trans = not all(abs(t) <= 1e-6 for t in trans)
    for i, r in enumerate(np.asarray(ranges, dtype=np.float32)):
        if float(r) > 0.0 and r < rmax and r > eps:
            theta = np.float32(np.float32(i) * ainc) + amin           # fp32 multiply, then fp32 add
            c, s = cos_sin(float(theta))
            x = float(r) * c
            y = float(r) * s
            if do_trans:
                x, y = transform_point(x, y, trans)
            out.append((x, y))
    return out


def transform_point(x, y, t):
    """core.h:28-31"""
    c, s = cos_sin(t[2])
    return x * c - y * s + t[0], x * s + y * c + t[1]


def cell_index(x, y, width, height, cs):
    """NDTFrame::getCellIndex, ndtframe.cpp:240-249"""
    W = int(math.ceil(width / cs))
    if x > -width / 2.0 and x < width / 2.0 and y > -height / 2.0 and y < height / 2.0:
        return int(math.floor((x + width / 2.0) / cs) + W * math.floor((y + height / 2.0) / cs))
    return -1


DBL_MIN = 2.2250738585072014e-308
DBL_EPS = 2.220446049250313e-16


def _givens(p, q):
    """JacobiRotation<double>::makeGivens, real case (Eigen/src/Jacobi/Jacobi.h)."""
    if q == 0.0:
        return (-1.0 if p < 0.0 else 1.0), 0.0
    if p == 0.0:
        return 0.0, (1.0 if q < 0.0 else -1.0)
    if abs(p) > abs(q):
        t = q / p
        u = math.sqrt(1.0 + t * t)
        if p < 0.0:
            u = -u
        c = 1.0 / u
        return c, -t * c
    t = p / q
    u = math.sqrt(1.0 + t * t)
    if q < 0.0:
        u = -u
    s = -1.0 / u
    return -t * s, s


def eigen_solver_2x2(m00, m01, m10, m11, scaled=True):
    """EigenSolver<Matrix2d>(M).pseudoEigenvalueMatrix().diagonal() (ndtcell.cpp:96-97), second restatement of Eigen
    3.3.7's RealSchur path (RealSchur::compute -> computeFromHessenberg -> splitOffTwoRows, EigenSolver::compute),
    written on a matrix of Python floats [[t00, t01], [t10, t11]] instead of the C oracle's flat array.
    scaled=False: without RealSchur::compute's scale / unscale (older 3.3.x)."""
    scale = 1.0
    if scaled:
        scale = max(abs(m00), abs(m01), abs(m10), abs(m11))
        if scale < DBL_MIN:
            return 0.0, 0.0
    t = [[m00 / scale, m01 / scale], [m10 / scale, m11 / scale]] if scaled else [[m00, m01], [m10, m11]]
    norm = abs(t[0][0]) + abs(t[1][0]) + abs(t[0][1]) + abs(t[1][1])
    if norm != 0.0:
        s = abs(t[0][0]) + abs(t[1][1])
        thr = max(s * DBL_EPS, DBL_MIN) if scaled else DBL_EPS * s
        if abs(t[1][0]) <= thr:
            t[1][0] = 0.0
        else:
            p = 0.5 * (t[0][0] - t[1][1])
            q = p * p + t[1][0] * t[0][1]
            if q >= 0.0:
                z = math.sqrt(abs(q))
                c, sn = _givens(p + z if p >= 0.0 else p - z, t[1][0])
                if not (c == 1.0 and sn == 0.0):
                    # applyOnTheLeft(0, 1, rot.adjoint()): rotation (c, -sn) on the two rows
                    for j in (0, 1):
                        x, y = t[0][j], t[1][j]
                        t[0][j] = c * x + (-sn) * y
                        t[1][j] = sn * x + c * y
                    # applyOnTheRight(0, 1, rot): rot.transpose() = (c, -sn) on the two columns
                    for i in (0, 1):
                        x, y = t[i][0], t[i][1]
                        t[i][0] = c * x + (-sn) * y
                        t[i][1] = sn * x + c * y
                t[1][0] = 0.0
    if scaled:
        t = [[v * scale for v in row] for row in t]
    if t[1][0] == 0.0:
        return t[0][0], t[1][1]
    re = t[1][1] + 0.5 * (t[0][0] - t[1][1])
    return re, re


def build_cells(points, width, height, cs):
    """Fresh-frame NDTCell::build + s_calc_covar_inverse, ndtcell.cpp:36-68,93-111.
    Returns {index: dict(count, built, mean, icov)}."""
    H = int(math.ceil(height / cs))
    W = int(math.ceil(width / cs))
    buckets = {}
    for (x, y) in points:
        k = cell_index(x, y, width, height, cs)
        if k != -1 and k < W * H:
            buckets.setdefault(k, []).append((x, y))
    cells = {}
    for k, pts in buckets.items():
        n = len(pts)
        sx = sy = 0.0
        for (x, y) in pts:
            sx += x
            sy += y
        cell = dict(count=n, built=n > 2, mean=None, icov=None)
        if n > 2:
            mx, my = sx / n, sy / n
            c00 = c01 = c10 = c11 = 0.0
            for (x, y) in pts:
                d0, d1 = x - mx, y - my
                c00 += d0 * d0
                c01 += d0 * d1
                c10 += d1 * d0
                c11 += d1 * d1
            c00, c01, c10, c11 = c00 / n, c01 / n, c10 / n, c11 / n
            ev = np.linalg.eigvalsh(np.array([[c00, c01], [c10, c11]]))   # LAPACK, not Eigen's RealSchur path
            large, small = max(ev), min(ev)
            det = 0.001 * large * large if small < 0.001 * large else c00 * c11 - c10 * c01
            cell["mean"] = (mx, my)
            cell["icov"] = (c11 / det, -c01 / det, -c10 / det, c00 / det)
        cells[k] = cell
    return cells


def cost(pose, cells, new_points, width, height, cs):
    """cost_function, core.cpp:26-48"""
    total = 0.0
    for (x, y) in new_points:
        qx, qy = transform_point(x, y, pose)
        k = cell_index(qx, qy, width, height, cs)
        c = cells.get(k)
        if k != -1 and c is not None and c["built"]:
            d0, d1 = qx - c["mean"][0], qy - c["mean"][1]
            a, b, cc, d = c["icov"]
            r0 = d0 * a + d1 * cc
            r1 = d0 * b + d1 * d
            total -= math.exp(-(r0 * d0 + r1 * d1) / 2.0)
    return total


def pso(guess, deviation, cells, new_points, width, height, cs, P, I, raw, w=0.8, c1=2.0, c2=2.0, wdamp=1.0):
    """pso_optimization, core.cpp:50-116, single-thread order; raw = rand() outputs."""
    it = iter(raw)
    f = lambda p: cost(p, cells, new_points, width, height, cs)   # noqa: E731

    def particle(mean, dev):
        pos = [mean[k] + (uniform_pm1(next(it)) * dev[k]) for k in range(3)]
        c = f(pos)
        return dict(pos=pos, vel=[0.0, 0.0, 0.0], bpos=list(pos), bcost=c)

    g = particle(guess, (1e-4, 1e-4, 1e-5))
    gpos, gcost = list(g["bpos"]), g["bcost"]
    ps = []
    for _ in range(P):
        p = particle(guess, deviation)
        ps.append(p)
        if p["bcost"] < gcost:
            gcost, gpos = p["bcost"], list(p["bpos"])
    for _ in range(I):
        for p in ps:
            for k in range(3):
                r1 = abs(uniform_pm1(next(it)))
                r2 = abs(uniform_pm1(next(it)))
                p["vel"][k] = w * p["vel"][k] + c1 * r1 * (p["bpos"][k] - p["pos"][k]) + c2 * r2 * (gpos[k] - p["pos"][k])
                p["pos"][k] = p["pos"][k] + p["vel"][k]
            c = f(p["pos"])
            if c < p["bcost"]:
                p["bcost"], p["bpos"] = c, list(p["pos"])
                if c < gcost:
                    gcost, gpos = c, list(p["pos"])
        w *= wdamp
    return np.array(gpos), gcost
