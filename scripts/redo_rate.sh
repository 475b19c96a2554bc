python scripts/replay_rate.py 200 > /dev/null 2>&1   # writes /tmp/scans.bin
for n in 1 2 3; do NDTPSO_RESIDENT=1 NDTPSO_SCORE=f32 host/replay/node_replay /tmp/scans.bin 60 0.5 50 30 7 2>&1 >/dev/null | tail -1; done
NDTPSO_LOG_REDO=1 NDTPSO_RESIDENT=1 NDTPSO_SCORE=f32 host/replay/node_replay /tmp/scans.bin 60 0.5 50 30 7 2>&1 >/dev/null | grep -c "handed to"
for n in 1 2; do NDTPSO_RESIDENT=1 NDTPSO_SCORE=f64 host/replay/node_replay /tmp/scans.bin 60 0.5 50 30 7 2>&1 >/dev/null | tail -1; done
