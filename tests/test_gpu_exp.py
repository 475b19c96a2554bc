"""The device arithmetic the fp64 score rests on.

1. exp_neg_half (ndtpso_kernels.hpp: the fp64 score's spelling of exp(-q / 2) -- the ROCm library's reduction, polynomial and
   ldexp without its two range guards, the factor -1/2 folded into the constants) against the library's own exp ON THE
   DEVICE, bit for bit: every binade of [-2^11, -2^-1074] and of (0, 2^10], > 10^8 arguments, plus the special values.
   NDTCell::normalDistribution (ndtcell.cpp:70-78) is what both evaluate.
2. The device library's exp and sincos against glibc's (the libm the reference runs on): mismatch RATES, reported -- the
   oracle is glibc, the fp64 mode is the device, and DESIGN.md used to say "equal in every case compared".
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ndtpso_slam_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def test_exp_neg_half_equals_the_library_exp_on_every_binade(ctx):
    # negative arguments: x = -(1 + u) 2^b, b = -1074 .. 10 covers [-2048, -4.9e-324]: everything from "exp = 1 - ulp"
    # over the gradual underflow of the result (x < -708) to +0. (x < -745.2) and the library's own guard (x < -1075)
    total = bad = 0
    for lo, hi, per in ((-1074, -1023, 20_000), (-1022, -61, 20_000), (-60, 10, 1_500_000)):
        n, b, x = ctx.selftest_exp(lo, hi, per, seed=lo & 0xffff)
        total += n
        bad += b
        assert b == 0, "exp_neg_half(-2x) != exp(x) for %d of %d arguments in binades [%d, %d], e.g. x = %r" % (b, n, lo, hi, x)
    # positive arguments up to 2^10 (an indefinite form's exponent; overflow to +inf above 709.78 is ldexp's in both)
    for lo, hi, per in ((-1074, -61, 5_000), (-60, 9, 200_000)):
        n, b, x = ctx.selftest_exp(lo, hi, per, seed=7, positive=True)
        total += n
        bad += b
        assert b == 0, "positive arguments: %d of %d differ in binades [%d, %d], e.g. x = %r" % (b, n, lo, hi, x)
    assert total > 100_000_000
    # the domain the kernels use it on ends at |x| < 2^40 (d64_cell_tame): up to there the result is +0. / +inf by ldexp alone
    for lo, hi in ((11, 39),):
        n, b, x = ctx.selftest_exp(lo, hi, 100_000, seed=3)
        assert b == 0, (b, n, x)
        n, b, x = ctx.selftest_exp(lo, hi, 100_000, seed=4, positive=True)
        assert b == 0, (b, n, x)


def test_exp_neg_half_special_values(ctx):
    thr = [-745.1332191019411, -745.1332191019412, -745.13321910194, -708.3964185322641, -709.782712893384,
           709.782712893384, 709.7827128933841, 709.78271289338397, 1024.0, -1075.0, -1074.9999999999998, -1075.0000000000002]
    x = np.array([0.0, -0.0, np.nan, -1e-320, 1e-320, -2.2250738585072014e-308, -1.0, -0.5, -37.0, -1e6, -1.8e6, -4.5e5,
                  -2.0 ** 39, 2.0 ** 39] + thr)
    x = np.concatenate([x, np.nextafter(thr, -np.inf), np.nextafter(thr, np.inf)])
    lib = ctx.device_math("exp", x)
    mine = ctx.device_math("exp_neg_half", x)
    same = (lib.view(np.uint64) == mine.view(np.uint64)) | (np.isnan(lib) & np.isnan(mine))
    assert same.all(), list(zip(x[~same], lib[~same], mine[~same]))
    assert np.isnan(mine[2]) and mine[0] == 1.0 and mine[9] == 0.0 and not np.signbit(mine[9])  # NaN stays, a miss adds +0.


def test_device_exp_and_sincos_against_glibc(ctx):
    """Reported, not required to be zero: the rates go to gpurun_out/ and DESIGN.md quotes them."""
    from oracle import pyoracle
    rng = np.random.default_rng(5)
    n = 10_000_000
    report = {}
    # exp on [-745, 0]: uniform, and log-uniform in |x| (the Gaussian terms of a score cluster near 0)
    for name, x in (("exp_uniform[-745,0]", -745.0 * rng.random(n)), ("exp_loguniform[-745,-1e-9]", -np.exp(rng.uniform(np.log(1e-9), np.log(745.0), n)))):
        d = ctx.device_math("exp", x)
        h = pyoracle.libm_exp(x)
        diff = d.view(np.int64) - h.view(np.int64)
        report[name] = {"n": n, "mismatch": int((diff != 0).sum()), "rate": float((diff != 0).mean()), "max_ulp": int(np.abs(diff).max())}
        assert np.abs(diff).max() <= 1, report
    x = rng.uniform(-np.pi, np.pi, n)
    ds, dc = ctx.device_math("sincos", x)
    hs, hc = pyoracle.libm_sincos(x)
    for nm, a, b in (("sin", ds, hs), ("cos", dc, hc)):
        diff = a.view(np.int64) - b.view(np.int64)
        report["sincos_%s_uniform[-pi,pi]" % nm] = {"n": n, "mismatch": int((diff != 0).sum()), "rate": float((diff != 0).mean()),
                                                    "max_ulp": int(np.abs(diff).max())}
        assert np.abs(diff).max() <= 2, report
    # the headings a PSO actually takes: |theta| < 0.05 rad around the guess
    x = rng.uniform(-0.05, 0.05, n)
    ds, dc = ctx.device_math("sincos", x)
    hs, hc = pyoracle.libm_sincos(x)
    for nm, a, b in (("sin", ds, hs), ("cos", dc, hc)):
        diff = a.view(np.int64) - b.view(np.int64)
        report["sincos_%s_uniform[-0.05,0.05]" % nm] = {"n": n, "mismatch": int((diff != 0).sum()), "rate": float((diff != 0).mean()),
                                                        "max_ulp": int(np.abs(diff).max())}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "device_vs_glibc.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
