"""The reference's real node translation units type-check, unchanged, against the drop-in headers.

Runs where /root/reference exists (the build container); on the GPU box, which has no reference tree, it skips.
Nothing of the reference is copied: scripts/node_syntax_check.py compiles the files in place with g++ -fsyntax-only
against host/include plus throw-away ROS / tf2 stand-in headers written into tmp_path."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "src", "ndtpso_slam_node.cpp"))
                                or shutil.which("g++") is None, reason="reference tree or g++ not present")


def _tool():
    spec = importlib.util.spec_from_file_location("node_syntax_check", os.path.join(ROOT, "scripts", "node_syntax_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_real_node_sources_compile_against_drop_in_headers(tmp_path):
    tool = _tool()
    for unit, rc, out in tool.check(REFERENCE, str(tmp_path)):
        assert rc == 0, "%s does not type-check against host/include:\n%s" % (unit, out)
        assert "warning" not in out, out
    # every ndtpso_slam/ header the node pulled in is the drop-in's, none the reference's own
    headers = tool.which_headers(REFERENCE, str(tmp_path))
    assert {os.path.basename(h) for h in headers} >= {"ndtframe.h", "ndtcell.h", "config.h"}
    assert all(h.startswith(os.path.join(ROOT, "host", "include")) for h in headers), headers


@pytest.mark.parametrize("method", ["align", "loadLaser", "update", "addPose", "dumpMap", "setTrans"])
def test_the_check_notices_a_missing_method(tmp_path, method):
    """Control: with one of the methods the node calls renamed away after the drop-in header has been read, the same
    compile must fail -- i.e. the green test above really exercises those call sites."""
    tool = _tool()
    poison = tmp_path / "poison.h"
    poison.write_text('#include "ndtpso_slam/ndtframe.h"\n#define %s %s_is_not_declared\n' % (method, method))
    results = tool.check(REFERENCE, str(tmp_path), extra_flags=("-include", str(poison)))
    unit, rc, out = results[0]
    assert rc != 0 and ("%s_is_not_declared" % method) in out
