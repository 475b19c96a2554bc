for v in arb2e-6 arb5e-6; do
export NDTPSO_LIB=ndtpso_slam_amd/lib/variants/$v.so
echo "== $v"
python scripts/r2_ab_modes.py 512 3 f32,exact 2>/dev/null
python scripts/r2_arb_stats.py 2>/dev/null | grep -A1 "^exact"
timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
done
