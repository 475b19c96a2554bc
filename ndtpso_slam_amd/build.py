"""Builds the gfx950 HIP library in-tree (ndtpso_slam_amd/lib/libndtpso_hip.so).

hipcc cross-compiles for gfx950 without a GPU present.  -ffp-contract=off is part of the
numerical contract: the fp64 PSO update / index arithmetic must round like the reference
(baseline x86-64, no FMA); fused multiply-adds exist only where the kernels spell fma().
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ndtpso_hip.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "ndtpso_kernels.hpp"), os.path.join(HERE, "csrc", "ndtpso_map.inc"),
        os.path.join(HERE, "csrc", "ndtpso_pairs_body.inc"),
        os.path.join(HERE, "csrc", "ndtpso_shard.inc"), os.path.join(HERE, "csrc", "ndtpso_selftest.inc"),
        os.path.join(os.path.dirname(HERE), "include", "ndtpso_hip.h")]
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libndtpso_hip.so")

# -mllvm -enable-ipra=0: without the interprocedural register allocation.  The exact mode's fp64 scores are two out-of-line
# functions (exact_unit_call / exact_task_call); with IPRA their callers allocate against the register set each callee was
# seen to write instead of the calling convention's.  Six builds of the round-4 sources made that way returned the same
# wrong poses from the arbitration's unit form on swarms kept in HBM (7 of 130 pairs, any workgroup size); the same six
# sources without IPRA, and every other build without it, are right (NOTEBOOK, "The unit form on swarms kept in HBM";
# tests/test_gpu_fullsize.py::test_unit_form_on_swarms_kept_in_hbm).  Cost: 4 % of the exact kernel when the switch was made,
# nothing measurable on the round's final kernels.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-enable-ipra=0", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


STAMP = LIB + ".sources"   # hash of the sources the library was built from (travels with the .so)


def _sources_hash() -> str:
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    """Content, not mtime, decides: a copy of the tree (the GPU box gets one) does not keep timestamps in order."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as f:
            return f.read().strip() != _sources_hash()
    except OSError:
        return True


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """Compile if the library is missing or older than its sources.  Safe when several processes call it at once (one
    rank per GPU importing the package): one of them compiles under a file lock, into a temporary name that is moved
    into place atomically; the others wait and find it fresh."""
    if force or needs_build():
        import fcntl
        os.makedirs(LIB_DIR, exist_ok=True)
        with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if force or needs_build():  # somebody else may have built it while this process waited
                    tmp = "%s.%d.tmp" % (LIB, os.getpid())
                    cmd = [hipcc()] + FLAGS + [SRC, "-o", tmp]
                    if verbose:
                        print(" ".join(cmd[:-1] + [LIB]))
                    try:
                        subprocess.check_call(cmd)
                        os.replace(tmp, LIB)
                        with open(STAMP, "w") as f:
                            f.write(_sources_hash() + "\n")
                    finally:
                        if os.path.exists(tmp):
                            os.remove(tmp)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


# Diagnostic builds of the same sources (never loaded by the product path; NDTPSO_LIB selects one for a process):
#   budget  -DNDTPSO_PHASE_BUDGET   per-workgroup, per-phase clocks (scripts/phase_budget.py)
#   verify  -DNDTPSO_VERIFY_MARGIN  every fp32 score checked against its fp64 value and an a-priori error bound
#                                   (scripts/verify_margin.py, tests/test_gpu_margin.py)
#   why     -DNDTPSO_WHY_BITS       status bits that say why an alignment was handed to the fp64-score kernel (scripts/shape_diag.py)
#   broken  -DNDTPSO_BREAK_ARBITRATION  the arbitration's unit form returns scores that are off by 2^-44: what the start-up check
#                                   of the exact mode exists to catch (tests/test_gpu_exact_check.py loads it in a subprocess)
VARIANTS = {"budget": ["-DNDTPSO_PHASE_BUDGET"], "verify": ["-DNDTPSO_VERIFY_MARGIN"], "broken": ["-DNDTPSO_BREAK_ARBITRATION"],
            "why": ["-DNDTPSO_WHY_BITS"]}


def variant_path(name: str) -> str:
    return os.path.join(LIB_DIR, "libndtpso_hip_%s.so" % name)


def build_variant(name: str, force: bool = False, verbose: bool = False) -> str:
    """Compile a diagnostic variant next to the shipped library (content-stamped like it)."""
    out, stamp = variant_path(name), variant_path(name) + ".sources"
    want = _sources_hash() + " " + " ".join(VARIANTS[name])
    fresh = False
    if os.path.exists(out) and not force:
        try:
            with open(stamp) as f:
                fresh = f.read().strip() == want
        except OSError:
            fresh = False
    if not fresh:
        os.makedirs(LIB_DIR, exist_ok=True)
        tmp = "%s.%d.tmp" % (out, os.getpid())
        cmd = [hipcc()] + FLAGS + VARIANTS[name] + [SRC, "-o", tmp]
        if verbose:
            print(" ".join(cmd[:-1] + [out]))
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, out)
            with open(stamp, "w") as f:
                f.write(want + "\n")
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] in VARIANTS:
        print(build_variant(sys.argv[1], force=True, verbose=True))
    else:
        print(build_hip(force=True, verbose=True))
