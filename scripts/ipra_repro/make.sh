#!/bin/bash
# Builds the reproducer of the code-generation defect that -mllvm -enable-ipra=0 in ndtpso_slam_amd/build.py works around.
# Needs the repository's history (commit b604a96: the round-4 sources that fail) and hipcc; no GPU.  Output: scripts/ipra_repro/_build/
#   tree/            that commit's python package, header and diagnostic script (its own capi: the C-ABI has grown since)
#   ipra1.so         its sources with the compiler's default interprocedural register allocation      -> 7 of 130 pairs wrong
#   ipra0.so         the same sources, -mllvm -enable-ipra=0                                          -> 130 of 130 right
#   ipra1_fenced.so  ipra1 with a barrier before and after every pass of the arbitration's units and the units' LDS scratch
#                    poisoned with NaN before every pass (a unit that was not written, or was read early, would surface as a NaN
#                    cost): a source-level race between the passes cannot survive this build
# Then, on a gfx950 box:  bash scripts/ipra_repro/run.sh
set -euo pipefail
cd "$(dirname "$0")/../.."
COMMIT=${1:-b604a96}
OUT=scripts/ipra_repro/_build
rm -rf "$OUT"; mkdir -p "$OUT/tree"
git archive "$COMMIT" ndtpso_slam_amd include scripts/units_hbm_diag.py | tar -x -C "$OUT/tree"
rm -rf "$OUT/tree/ndtpso_slam_amd/lib"
SRC="$OUT/tree/ndtpso_slam_amd/csrc/ndtpso_hip.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -Wall -Wno-unused-function"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC $FLAGS "$SRC" -o "$OUT/ipra1.so" &
$HIPCC $FLAGS -mllvm -enable-ipra=0 "$SRC" -o "$OUT/ipra0.so" &
# the fenced variant: a patched copy of the kernels
mkdir -p "$OUT/fenced"; cp -r "$OUT/tree/ndtpso_slam_amd" "$OUT/tree/include" "$OUT/fenced/"
python3 - "$OUT/fenced/ndtpso_slam_amd/csrc/ndtpso_kernels.hpp" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "    if (4 * t0 + wave_id() < 4 * t1) exact_units<BYTE, INL>(ap, 4 * t0 + wave_id(), n_waves, 4 * t1);\n    __syncthreads();\n"
new = ("    {  // ipra_repro: poison the units' scratch, fence the pass on both sides\n"
       "      typedef double __attribute__((address_space(3))) * lds_d_t;\n"
       "      __syncthreads();\n"
       "      for (int q = threadIdx.x; q < ap->xs_slots * kWave; q += blockDim.x) *(lds_d_t)(uintptr_t)(ap->xs_lds + (unsigned)q * 8u) = __builtin_nan(\"\");\n"
       "      __syncthreads();\n"
       "    }\n"
       "    if (4 * t0 + wave_id() < 4 * t1) exact_units<BYTE, INL>(ap, 4 * t0 + wave_id(), n_waves, 4 * t1);\n"
       "    __syncthreads();\n    __threadfence_block();\n    __syncthreads();\n")
assert s.count(old) == 1
open(p, "w").write(s.replace(old, new))
PY
$HIPCC $FLAGS "$OUT/fenced/ndtpso_slam_amd/csrc/ndtpso_hip.hip" -o "$OUT/ipra1_fenced.so" &
wait
rm -rf "$OUT/fenced"
$HIPCC --version | head -2 > "$OUT/hipcc_version.txt"
ls -la "$OUT"
