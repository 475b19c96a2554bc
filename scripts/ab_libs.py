"""Same-box A/B of library builds: bench.py (config 3, short) run alternately with NDTPSO_LIB = each library, several
rounds, so that box-to-box spread (+-3 %) does not decide.   python scripts/ab_libs.py [--rounds 3] [--score exact,f32] a.so b.so ...
(GPU box; the libraries travel in the tree, e.g. under ab_prev/)"""
import argparse, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--score", default="exact")
ap.add_argument("--steps", type=int, default=120)
args = ap.parse_args()
res = {}
for r in range(args.rounds):
    for lib in args.libs:
        for score in args.score.split(","):
            env = dict(os.environ, NDTPSO_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", "10", "--cpu-sample", "0",
                                  "--no-latency", "--score", score], env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                res.setdefault((lib, score), []).append((d["value"], d["value_one_batch_at_a_time"], d["roofline"]["kernel_ms"]))
            except Exception as e:  # noqa: BLE001
                print("failed:", lib, score, out.stderr[-400:])
for (lib, score), v in res.items():
    a = np.array(v)
    print("%-40s %-6s two-in-flight %8.0f  one-at-a-time %8.0f  kernel %.4f ms   (runs: %s)"
          % (os.path.basename(lib), score, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), " ".join("%.0f" % x for x in a[:, 0])))
