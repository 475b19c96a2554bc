python scripts/r2_ab_modes.py 512 3 f32,exact,f64 2>/dev/null | grep "ms/launch"
python scripts/config5_profile.py 256 f32 2 2>/dev/null | tail -1
python scripts/config5_profile.py 256 exact 2 2>/dev/null | tail -1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
