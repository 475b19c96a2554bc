"""How often the exact mode arbitrates on BASELINE config 3 (512 pairs, 70 x 70): comparisons arbitrated per alignment."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ndtpso_slam_amd import capi, synth
p = synth.make_pairs(512, seed=2024)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
ctx = capi.Context(0)
for mode, name in ((capi.SCORE_F32, "f32"), (capi.SCORE_EXACT, "exact"), (capi.SCORE_F64, "f64")):
    for _ in range(3):
        t0 = time.perf_counter()
        pose, cost, st = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (0.1, 0.1, 3.1415e-3),
                                         capi.PSOConfig.make(70, 70), seeds=p.seeds, mode=mode)
        dt = time.perf_counter() - t0
    a = st["arbitrated"]
    print(name, "%.2f ms (host buffers)" % (1e3 * dt), "arbitrated: total", int(a.sum()), "mean %.2f" % a.mean(), "max", int(a.max()),
          "alignments with any:", int((a > 0).sum()), "status", np.unique(st["status"]),
          "wg time (us): mean %.0f max %.0f" % ((st["t_end"] - st["t_start"]).astype(np.uint32).mean() / 100, (st["t_end"] - st["t_start"]).astype(np.uint32).max() / 100))
    dur = (st["t_end"] - st["t_start"]).astype(np.uint32) / 100.0
    end = (st["t_end"] - st["t_start"].min()).astype(np.uint32) / 100.0
    print("   wg duration us: p50 %.0f p90 %.0f p99 %.0f max %.0f; finish time: p50 %.0f p99 %.0f max %.0f; corr(dur, arbitrated) %.2f, corr(dur, evals) %.2f"
          % (np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), np.percentile(end, 50), np.percentile(end, 99), end.max(),
             np.corrcoef(dur, st["arbitrated"] + 1e-9 * np.arange(len(dur)))[0, 1], np.corrcoef(dur, st["cost_evals"])[0, 1]))
    if name == "exact":
        for k in range(0, 15):
            sel = a == k
            if sel.any(): print("     arbitrated %2d: %3d alignments, mean duration %.0f us" % (k, sel.sum(), dur[sel].mean()))
