run() { python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', round(d['value']), round(d['roofline']['kernel_ms'],3))"; }
V=$PWD/ndtpso_slam_amd/lib/variants
for W in 8 9 10; do TAG="mw4_u2 w$W" NDTPSO_LIB=$V/mw4_u2.so NDTPSO_WAVES=$W run; done
for W in 8 10; do TAG="mw5_u2 w$W" NDTPSO_LIB=$V/mw5_u2.so NDTPSO_WAVES=$W run; done
for W in 10; do TAG="mw5_u4 w$W" NDTPSO_LIB=$V/mw5_u4.so NDTPSO_WAVES=$W run; done
for W in 10 12; do TAG="mw6_u2 w$W" NDTPSO_LIB=$V/mw6_u2.so NDTPSO_WAVES=$W run; done
for W in 12; do TAG="mw6_u4 w$W" NDTPSO_LIB=$V/mw6_u4.so NDTPSO_WAVES=$W run; done
for W in 14 16; do TAG="mw8_u2 w$W" NDTPSO_LIB=$V/mw8_u2.so NDTPSO_WAVES=$W run; done
for W in 16; do TAG="mw8_u4 w$W" NDTPSO_LIB=$V/mw8_u4.so NDTPSO_WAVES=$W run; done
