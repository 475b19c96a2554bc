"""CPU checks of the C-ABI library: it builds/loads, exports every symbol include/ndtpso_hip.h declares,
struct layouts match the binding, and without a HIP device it fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ndtpso_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ndtpso_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ndtpso_slam_amd import capi
    L = capi.load()
    names = _declared()
    assert len(names) >= 16
    for n in names:
        assert hasattr(L, n), n
    assert sorted(capi.EXPORTS) == names


def test_struct_layouts():
    from ndtpso_slam_amd import capi
    assert C.sizeof(capi.PSOConfig) == 48       # 3 x int32 + pad + 4 x double (config.h:27-38)
    assert C.sizeof(capi.Grid) == 16
    assert C.sizeof(capi.ScanGeom) == 20
    assert C.sizeof(capi.CellRow) == 64
    assert C.sizeof(capi.AlignStats) == 32


def test_rand_draws_and_footprint_without_gpu():
    from ndtpso_slam_amd import capi
    L = capi.load()
    cfg = capi.PSOConfig.make(70, 70)
    assert L.ndtpso_rand_draws(C.byref(cfg)) == 29613
    geom = capi.ScanGeom(1081, -2.356194, 4.712389 / 1080, 30.0, 0.1)
    rc, lds, thr = capi.align_pairs_footprint(geom, capi.Grid(60, 60, 0.5), cfg)
    assert rc == 0 and 0 < lds <= 80 * 1024 and thr % 64 == 0      # two workgroups per CU
    rc, lds, thr = capi.align_pairs_footprint(geom, capi.Grid(60, 60, 0.3), cfg)   # non power-of-two cells
    assert rc == 0 and 0 < lds <= 160 * 1024
    # BASELINE config 5 (2048 particles, 2048 beams, 0.25 m cells): table + points in LDS, swarm in HBM, 1 WG/CU
    big = capi.PSOConfig.make(200, 2048)
    rc, lds, thr = capi.align_pairs_footprint(capi.ScanGeom(2048, -2.3, 0.002, 30.0, 0.1), capi.Grid(60, 60, 0.25), big)
    assert rc == 0 and 80 * 1024 < lds <= 160 * 1024 and thr == 1024
    # a table that must shrink to leave two workgroups per CU is sized to the cell: one more row and column would not fit
    # (in steps of four cells a side it ended 4 % short of a room's box at 0.25 m cells -- the same in the 60 and the 100 m frame)
    plans = []
    for frame in (60, 100):
        rc, plan = capi.align_pairs_describe(geom, capi.Grid(frame, frame, 0.25), cfg, capi.SCORE_EXACT, 512)
        assert rc == 0 and plan["workgroups_per_cu"] == 2 and plan["table_form"] == 2
        side = int((plan["table_bytes"] // 2) ** 0.5)          # u16 entries, (side + 1)^2 of them + the records
        assert 80 * 1024 - plan["lds_bytes"] < 2 * (2 * side + 3) + 16
        plans.append((plan["lds_bytes"], plan["table_bytes"]))
    assert plans[0] == plans[1]
    # a scan whose points alone exceed LDS is refused, loudly
    rc, lds, _ = capi.align_pairs_footprint(capi.ScanGeom(12000, -2.3, 0.0004, 30.0, 0.1), capi.Grid(60, 60, 0.25), big)
    assert rc == capi.E_CAPACITY and lds == 0
    # a frame whose cell counts would wrap the reference's uint16_t (ndtframe.h:32), or a degenerate one, is refused
    for bad in (capi.Grid(60000, 60, 0.5), capi.Grid(60, 60, 0.0), capi.Grid(0, 60, 0.5)):
        rc, _, _ = capi.align_pairs_footprint(geom, bad, cfg)
        assert rc == capi.E_ARG


def test_no_cpu_fallback():
    """Without a usable HIP device the product path refuses to run."""
    import torch
    from ndtpso_slam_amd import capi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.NdtpsoError):
        capi.Context(0)


def test_product_package_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under ndtpso_slam_amd/, include/, host/ or scripts/ (the measurement
    tools) may reference it; the campaigns that check against it live under tests/."""
    bad = []
    for base in ("ndtpso_slam_amd", "include", "host", "scripts"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp", ".c", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"pyoracle|ndtpso_oracle|orc_[a-z_]+\(", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
