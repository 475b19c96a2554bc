python -m pytest tests -m gpu -x -q -s 2>&1 | tail -12
run() { python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']; print('$TAG', round(d['value']), round(d['roofline']['kernel_ms'],3), e['mean_cost_evals_per_alignment'], e['cost_evals_min_max'], e['rounds_min_max'])"; }
for G in 1 2 3 4; do TAG="dense G$G" NDTPSO_GROUP=$G run; done
TAG="dense G2 identical" run --identical
TAG="dense G2 w16" NDTPSO_WAVES=16 run
TAG="bitmap G2" NDTPSO_PATH=1 run
TAG="f64 G2" run --score f64
TAG="dense 2048 pairs" run --pairs 2048
python bench.py --steps 10 --warmup 2 --cpu-sample 64 2>/dev/null | tail -1
