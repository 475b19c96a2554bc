python -m pytest tests -m gpu -x -q -s 2>&1 | tail -15
run() { python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', round(d['value']), round(d['roofline']['kernel_ms'],3), d['extra'].get('parity_sample_max_abs_dpose'))"; }
for W in 7 8 14 16; do TAG="dense w$W" NDTPSO_WAVES=$W run; done
for W in 8 14; do TAG="bitmap w$W" NDTPSO_PATH=1 NDTPSO_WAVES=$W run; done
TAG="f64 w8" NDTPSO_WAVES=8 run --score f64
TAG="f64 w14" NDTPSO_WAVES=14 run --score f64
python bench.py --steps 5 --warmup 1 --cpu-sample 64 2>/dev/null | tail -1
