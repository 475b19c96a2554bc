"""Is a kernel's machine code the same in two builds of the library?  Disassembles ONE kernel (by a substring of its demangled
name) out of each library's gfx950 code object and compares the instruction streams (addresses and symbol offsets aside).

    python scripts/kernel_isa_diff.py a.so b.so "k_align_pairs<0, 3, false, true, true, 0, false>" [name in b.so, if it differs]

(round 6's last library names the kernels with an eighth template argument -- "k_align_pairs<0, 3, false, true, true, 0, false, false>" --
and its benchmark kernels differ from round 5's by 80 of 11 176 instructions since the two-items-per-wave twins exist: NOTEBOOK.)
"""
import difflib
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix  # noqa: E402


def kernel_text(lib, want):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(isa_mix.extract_code_object(lib))
        f.flush()
        nm = subprocess.check_output([os.path.join(isa_mix.LLVM, "llvm-readelf"), "--symbols", "--wide", f.name], text=True)
        syms = [l.split()[-1] for l in nm.splitlines() if " FUNC " in l]
        hit = None
        for sname in syms:
            dem = subprocess.run(["c++filt", sname], capture_output=True, text=True).stdout.strip()
            if want.replace(" ", "") in dem.replace(" ", "") and not sname.endswith(".kd"):
                hit = sname
                break
        if not hit:
            raise SystemExit("%s: no kernel matching %r" % (lib, want))
        txt = subprocess.check_output([os.path.join(isa_mix.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr",
                                       "--disassemble-symbols=" + hit, f.name], text=True)
    out = []
    for l in txt.splitlines():
        l = l.split("//")[0].strip()
        if not l or l.endswith(":") or l.startswith("Disassembly") or "file format" in l:
            continue
        out.append(re.sub(r"<[^>]*>", "<sym>", l))
    return out


def main():
    a, b, want = sys.argv[1], sys.argv[2], sys.argv[3]
    want_b = sys.argv[4] if len(sys.argv) > 4 else want   # (a template parameter that changed its type prints differently)
    ta, tb = kernel_text(a, want), kernel_text(b, want_b)
    print("%s: %d instructions   %s: %d instructions" % (os.path.basename(a), len(ta), os.path.basename(b), len(tb)))
    if ta == tb:
        print("identical")
        return
    d = [l for l in difflib.unified_diff(ta, tb, lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---")]
    print("%d differing lines" % len(d))
    for l in d[:40]:
        print(l)


if __name__ == "__main__":
    main()
