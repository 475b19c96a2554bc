# the rows of DESIGN.md section 5 that come from bench.py variants (each under a timeout)
run() { timeout 120 python bench.py --cpu-sample 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']), 'align/s', round(d['roofline']['kernel_ms'],3), 'ms')"; }
run
run --score f64
run --pairs 2048 --steps 5
run --pairs 4096 --steps 3
run --pairs 256
run --pairs 1024 --steps 5
