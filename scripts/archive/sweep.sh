run() { python bench.py --steps 5 --warmup 1 --cpu-sample 32 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']; print('$TAG', round(d['value']), round(d['roofline']['kernel_ms'],3), e['parity_sample_max_abs_dpose'])"; }
V=$PWD/ndtpso_slam_amd/lib/variants
TAG="pipe u4 w8" run
TAG="nopipe u4 w8" NDTPSO_LIB=$V/nopipe_u4.so run
TAG="pipe u2 w8" NDTPSO_LIB=$V/pipe_u2.so run
TAG="pipe u1 w8" NDTPSO_LIB=$V/pipe_u1.so run
TAG="pipe u2 mw5 w8" NDTPSO_LIB=$V/pipe_u2_mw5.so run
TAG="pipe u2 w8 G1" NDTPSO_GROUP=1 NDTPSO_LIB=$V/pipe_u2.so run
