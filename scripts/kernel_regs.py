"""Register use of the kernels in the shipped library (code object metadata): VGPRs, spills, scratch, SGPRs.

    python scripts/kernel_regs.py [substring of the kernel name, default k_align_pairs]
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "k_align_pairs"
    lib = os.environ.get("NDTPSO_LIB", os.path.join(ROOT, "ndtpso_slam_amd", "lib", "libndtpso_hip.so"))
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(isa_mix.extract_code_object(lib))
        f.flush()
        out = subprocess.check_output([os.path.join(isa_mix.LLVM, "llvm-readelf"), "--notes", f.name], text=True)
    for b in out.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        if want not in dem:
            continue
        get = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)  # noqa: E731
        print("%-60s vgpr %s spill %s scratch %s sgpr %s lds_static %s" % (dem.replace("void ndtpso::", ""), get("vgpr_count"), get("vgpr_spill_count"),
                                                             get("private_segment_fixed_size"), get("sgpr_count"), get("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
