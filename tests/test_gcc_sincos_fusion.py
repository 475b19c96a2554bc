"""The oracle takes the cosine and sine of a pose's heading from ONE glibc sincos() call (oracle/ndtpso_oracle.c: ref_cos_sin), and
so do the host side's beam directions -- because that is what GCC makes of the reference's source: transform_point
(include/ndtpso_slam/core.h:28-31) and laser_to_point (:45-47) write cos(t) and sin(t) of the same argument side by side, and
GCC at -O3 (the reference's flags, CMakeLists.txt:5-9: -std=c++14, Release = -O3) fuses such a pair into a call of sincos.
glibc's sincos is not bit-identical to its cos / sin for ~0.15 % of arguments, so the assumption carries weight: here it is
checked on this image's GCC -- the two expressions, in the reference's shape, compiled with the reference's flags, and the
object code inspected."""
import os
import re
import subprocess

SRC = r"""
#include <cmath>
#include <vector>
using namespace std;
struct V2 { double a, b; double x() const { return a; } double y() const { return b; } };
struct V3 { double a, b, c; double x() const { return a; } double y() const { return b; } double z() const { return c; } };
// the shape of transform_point: both functions of trans.z() in one braced initialiser
V2 shape_transform_point(const V2 &point, const V3 &trans) {
  return {point.x() * cos(trans.z()) - point.y() * sin(trans.z()) + trans.x(),
          point.x() * sin(trans.z()) + point.y() * cos(trans.z()) + trans.y()};
}
// the shape of laser_to_point: float arguments widened at the call
V2 shape_laser_to_point(float r, float theta) {
  return {double(r) * cos(double(theta)), double(r) * sin(double(theta))};
}
"""


def _calls(obj, fn):
    out = subprocess.check_output(["objdump", "-dr", "--no-show-raw-insn", "-C", obj], text=True)
    m = re.search(r"^[0-9a-f]+ <%s\(.*?\)>:\n(.*?)(?:\n\n|\Z)" % re.escape(fn), out, re.S | re.M)
    assert m, out[:500]
    return re.findall(r"R_X86_64_PLT32\s+(\w+)", m.group(1)) + re.findall(r"call\s+\S+ <(\w+)@plt>", m.group(1))


def test_gcc_fuses_cos_and_sin_of_one_argument_into_sincos(tmp_path):
    src, obj = tmp_path / "shape.cpp", tmp_path / "shape.o"
    src.write_text(SRC)
    subprocess.check_call(["g++", "-std=c++14", "-O3", "-Wall", "-Wextra", "-c", str(src), "-o", str(obj)])
    for fn in ("shape_transform_point", "shape_laser_to_point"):
        calls = _calls(str(obj), fn)
        assert calls == ["sincos"], (fn, calls)     # one call, and it is sincos: no cos, no sin
    # ... and without optimisation (not how the reference is built) the pair stays two calls each: the fusion is the optimiser's
    obj0 = tmp_path / "shape0.o"
    subprocess.check_call(["g++", "-std=c++14", "-O0", "-c", str(src), "-o", str(obj0)])
    libm = [c for c in _calls(str(obj0), "shape_transform_point") if c in ("cos", "sin", "sincos")]
    assert sorted(set(libm)) == ["cos", "sin"] and len(libm) == 4, libm


def test_the_oracle_and_the_host_side_spell_sincos():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "sincos(th, s, c)" in open(os.path.join(root, "oracle", "ndtpso_oracle.c")).read()
    assert "::sincos(" in open(os.path.join(root, "ndtpso_slam_amd", "csrc", "ndtpso_hip.hip")).read()
