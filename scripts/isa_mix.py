"""Instruction mix of the score loop of the fused pairs kernel, from the library that is actually shipped, and the issue-time
floor that follows from it -- the `peak` of bench.py's roofline.

    python scripts/isa_mix.py [--lib ndtpso_slam_amd/lib/libndtpso_hip.so] [--kernel "k_align_pairs<0, 3, false, true, true, 0, false>"]
                              [--ubench profiles/r06_ubench_valu.txt] [--out profiles/r06_isa_mix.json]
                              [--dump profiles/r06_score_loop_isa.txt]

What it does
  1. pulls the gfx950 code object out of the library's .hip_fatbin section (clang offload bundle) and disassembles it
     with llvm-objdump;
  2. finds, inside the named kernel, the steady-state trip of the score loop of the PSO iterations: the innermost loops
     with four `v_exp_f32` in a row (four 64-point chunks in flight, score_trip_dense<4>); of the copies the compiler
     makes (frame-clip variant or not, swarm initialisation or iteration) it takes the shortest -- the no-clip
     iteration loop the benchmark's grid runs;
  3. counts the loop body's instructions per mnemonic and prices every VALU instruction with the issue time measured by
     scripts/ubench_valu.hip on the same chip (all CUs busy, 4 waves per SIMD, independent chains: time per wave64
     instruction per SIMD).  Mnemonics the microbenchmark does not cover are priced by encoding class (VOP2 / VOP3 /
     fp64 / transcendental) and listed under "assumed".
Floor: a SIMD issues one VALU instruction at a time, so the loop cannot run faster than
    sum over VALU instructions (issue time) per trip  /  4 chunks,
whatever the LDS and the other waves do.  bench.py multiplies by the chunks a launch scores and divides by the 1024
SIMDs: that is `roofline.peak` (as a rate) -- it ignores everything outside the loop (pose constants, the five-chunk
tail trip, reductions, the PSO itself), so it is a true lower bound of the kernel time.
(Since the no-clamp form of round 2 a 17-chunk list -- 1081 beams -- is scored in straight-line trips of 6 + 6 + 5 chunks
with the same per-chunk instruction mix; the loop found here is that form's four-chunk loop, which every other list
length runs.)
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def extract_code_object(lib: str) -> bytes:
    out = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", lib], text=True)
    m = re.search(r"\.hip_fatbin\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", out)
    if not m:
        raise SystemExit("no .hip_fatbin section in " + lib)
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    with open(lib, "rb") as f:
        f.seek(off)
        data = f.read(size)
    if data[:24] != b"__CLANG_OFFLOAD_BUNDLE__":
        raise SystemExit("unexpected fat binary format (compressed bundle?)")
    n = struct.unpack_from("<Q", data, 24)[0]
    p = 32
    for _ in range(n):
        o, s, t = struct.unpack_from("<QQQ", data, p)
        p += 24
        triple = data[p:p + t].decode()
        p += t
        if "gfx950" in triple and s:
            return data[o:o + s]
    raise SystemExit("no gfx950 code object in the bundle")


def disassemble(code: bytes) -> list[str]:
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(code)
        path = f.name
    try:
        txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", "--no-show-raw-insn",
                                       "--no-leading-addr", path], text=True)
    finally:
        os.remove(path)
    return txt.splitlines()


def kernel_body(lines: list[str], kernel: str) -> list[str]:
    want = kernel.replace(" ", "")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^[0-9a-f]* ?<(.*)>:$", l.strip())
        if m and m.group(1).replace(" ", "").startswith("void" + want + "(") or (m and m.group(1).replace(" ", "").startswith(want + "(")):
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found: " + kernel)
    body = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]* ?<[^>]*>:$", l.strip()) and not l.strip().startswith("<L"):
            break
        body.append(l)
    return body


def _addr(line: str):
    m = re.search(r"//\s*([0-9A-Fa-f]{8,16}):", line)
    return int(m.group(1), 16) if m else None


def find_trips(body: list[str], marker: str = "v_exp_f32", span: int = 8):
    """Innermost loops containing four consecutive v_exp_f32: [(first line, last line)], by their backward branch
    (llvm-objdump prints a branch as a signed word offset and every instruction's address in its trailing comment)."""
    ins = [(i, l.strip(), _addr(l)) for i, l in enumerate(body) if l.strip() and _addr(l) is not None]
    by_addr = {a: i for i, _, a in ins}
    exp_pos = [k for k, (_, t, _) in enumerate(ins) if t.startswith(marker)]
    trips, k = [], 0
    while k + 3 < len(exp_pos):
        if exp_pos[k + 3] - exp_pos[k] <= span:   # four in a row (a few scalar instructions may sit between them)
            first = ins[exp_pos[k]][2]
            for q in range(exp_pos[k + 3] + 1, min(len(ins), exp_pos[k + 3] + 80 + span)):
                i, t, a = ins[q]
                m = re.match(r"s_cbranch_\w+\s+(\d+)", t)
                if m:
                    off = int(m.group(1))
                    off = off - 65536 if off >= 32768 else off
                    target = a + 4 + 4 * off
                    if target < first and target in by_addr and a - target < 4096:
                        trips.append((by_addr[target], i))
                        break
                if t.startswith("s_endpgm"):
                    break
            k += 4
        else:
            k += 1
    return trips


def parse_ubench(path: str):
    """name -> ns per wave64 instruction per SIMD (scripts/ubench_valu.hip output)."""
    base_ns, rel = None, {}
    for l in open(path):
        m = re.match(r"v_fma_f32: ([0-9.]+) ns", l)
        if m:
            base_ns = float(m.group(1))
        m = re.match(r"(\S+)\s+([0-9.]+) cyc", l)
        if m:
            rel[m.group(1)] = float(m.group(2))
        m = re.match(r"cvt_f32_f64 \+ cvt_f64_f32\s+([0-9.]+) cyc", l)
        if m:
            rel["v_cvt_f32_f64"] = rel["v_cvt_f64_f32"] = float(m.group(1)) / 2
        m = re.match(r"cvt_i32_f64 \+ cvt_f64_i32\s+([0-9.]+) cyc", l)
        if m:
            rel["v_cvt_i32_f64"] = rel["v_cvt_f64_i32"] = float(m.group(1)) / 2
    if base_ns is None:
        raise SystemExit("no v_fma_f32 line in " + path)
    # the microbenchmark prints times relative to v_fma_f32 = 2 "cycles": ns = value / 2 * base
    return {k: v / 2.0 * base_ns for k, v in rel.items()}, base_ns


# microbenchmark kernel name -> mnemonic(s) it prices
UBENCH_NAMES = {
    "k_fma64": ["v_fma_f64", "v_fmac_f64"], "k_add64": ["v_add_f64"], "k_mul64": ["v_mul_f64"], "k_cmp64": ["v_cmp_lt_f64", "v_cmp_gt_f64"],
    "k_pkfma32": ["v_pk_fma_f32"], "k_pkmul32": ["v_pk_mul_f32", "v_pk_add_f32"], "k_mul32": ["v_mul_f32"], "k_add32": ["v_add_f32", "v_sub_f32"],
    "k_exp32": ["v_exp_f32"], "k_addu": ["v_add_u32", "v_sub_u32"], "k_and": ["v_and_b32", "v_or_b32"], "k_lshr": ["v_lshrrev_b32", "v_lshlrev_b32"],
    "k_mad24": ["v_mad_u32_u24"], "k_cndm": [], "k_mov": ["v_mov_b32"], "k_add3": ["v_add3_u32", "v_add_lshl_u32"],
    "k_lshladd": ["v_lshl_add_u32"], "k_fmac32": ["v_fmac_f32"], "k_minu32": ["v_min_u32"], "k_mulu24": ["v_mul_u32_u24"],
    "k_pkadd32": ["v_pk_add_f32"], "k_addlshl": ["v_add_lshl_u32"], "k_cndmask": ["v_cndmask_b32"], "k_fma32": ["v_fma_f32"],
}


def price_table(ub: dict, base_ns: float):
    t = {"v_fma_f32": base_ns}
    for k, names in UBENCH_NAMES.items():
        if k in ub:
            for n in names:
                t[n] = ub[k]
    for k, v in ub.items():
        if k.startswith("v_"):
            t[k] = v
    return t


def classify(mn: str) -> str:
    if mn.startswith("v_exp") or mn.startswith("v_rcp") or mn.startswith("v_sqrt") or mn.startswith("v_log"):
        return "transcendental"
    if "f64" in mn:
        return "fp64"
    return "vop"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "ndtpso_slam_amd", "lib", "libndtpso_hip.so"))
    ap.add_argument("--kernel", default="k_align_pairs<0, 3, false, true, true, 0, false>")
    ap.add_argument("--ubench", default=os.path.join(ROOT, "profiles", "r06_ubench_valu.txt"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_isa_mix.json"))
    ap.add_argument("--dump", default=os.path.join(ROOT, "profiles", "r06_score_loop_isa.txt"))
    ap.add_argument("--marker", default="v_exp_f32", help="the instruction a four-chunk trip holds four of (fp64-score kernels: v_ldexp_f64, "
                    "the tail of the library exp)")
    ap.add_argument("--marker-span", type=int, default=8, help="the four lie within this many instructions of each other (fp64: 60)")
    args = ap.parse_args()

    lines = disassemble(extract_code_object(args.lib))
    body = kernel_body(lines, args.kernel)
    trips = find_trips(body, args.marker, args.marker_span)
    if not trips:
        raise SystemExit("no four-chunk score trip found in " + args.kernel)
    def insns(a, b):
        return [re.sub(r"\s*//.*$", "", l.strip()) for l in body[a:b + 1] if l.strip() and _addr(l) is not None]
    sizes = [len(insns(a, b)) for a, b in trips]
    pick = min(range(len(trips)), key=lambda i: sizes[i])
    a, b = trips[pick]
    loop = insns(a, b)
    mn = [re.split(r"\s+", l)[0] for l in loop]
    mn = [re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", m) for m in mn]
    counts = collections.Counter(mn)
    ub, base_ns = parse_ubench(args.ubench)
    price = price_table(ub, base_ns)
    cheap = price.get("v_mul_f32", base_ns)
    fallback = {"vop": price.get("v_add3_u32", 2 * cheap), "fp64": price.get("v_fma_f64", 2 * cheap), "transcendental": price.get("v_exp_f32", 3 * cheap)}
    valu_ns, assumed, rows = 0.0, {}, []
    for m, c in sorted(counts.items()):
        if not m.startswith("v_"):
            continue
        if m in price:
            ns = price[m]
        else:
            ns = fallback[classify(m)]
            assumed[m] = ns
        valu_ns += c * ns
        rows.append({"mnemonic": m, "count": c, "issue_ns": round(ns, 4)})
    n_valu = sum(c for m, c in counts.items() if m.startswith("v_"))
    out = {
        "kernel": args.kernel,
        "library": os.path.relpath(args.lib, ROOT),
        "loop": "steady-state trip of the score loop (4 chunks of 64 points in flight), shortest of %d copies (%s instructions)" % (len(trips), sizes),
        "instructions_per_trip": len(loop),
        "valu_per_trip": n_valu, "lds_per_trip": sum(c for m, c in counts.items() if m.startswith("ds_")),
        "salu_per_trip": sum(c for m, c in counts.items() if m.startswith("s_")),
        "valu_per_chunk": n_valu / 4.0,
        "valu_issue_ns_per_chunk": valu_ns / 4.0,
        "ubench": {"file": os.path.relpath(args.ubench, ROOT), "v_fma_f32_ns": base_ns},
        "valu": rows, "assumed_prices": assumed,
        "other": {m: c for m, c in sorted(counts.items()) if not m.startswith("v_")},
    }
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    with open(args.dump, "w") as f:
        f.write("; %s -- steady-state score trip (4 x 64 points), llvm-objdump of %s\n" % (args.kernel, out["library"]))
        f.write("\n".join(loop) + "\n")
    print(json.dumps({k: out[k] for k in ("loop", "instructions_per_trip", "valu_per_trip", "lds_per_trip", "valu_per_chunk",
                                          "valu_issue_ns_per_chunk", "assumed_prices")}, indent=1))


if __name__ == "__main__":
    main()
