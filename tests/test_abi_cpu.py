"""The C-ABI library must load without a GPU and export every symbol include/livo2_hip.h declares; without a device it must fail
loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest


def _declared_symbols(header_path):
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(livo2_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(livo2):
    names = _declared_symbols(livo2.abi.HEADER_PATH)
    assert len(names) >= 20
    lib = livo2.abi.load_library()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/livo2_hip.h but not exported by liblivo2_hip.so"
    # and the Python binding covers exactly the declared surface
    assert sorted(livo2.abi.SIGNATURES) == names


def test_struct_layouts_match_header(livo2):
    a = livo2.abi
    assert C.sizeof(a.State) == 8 * (9 + 3 + 1 + 12 + 361)
    assert C.sizeof(a.LidarSums) == 8 * 43 + 8
    assert C.sizeof(a.LidarResult) == C.sizeof(a.State) + 8 + a.MAX_ITERS * C.sizeof(a.LidarSums) + a.MAX_ITERS * 19 * 8 + 24
    assert C.sizeof(a.VisualStep) == 24 + 8 * (49 + 7 + 19)
    assert C.sizeof(a.LidarCfg) == 8 + 8 * 5 + 8 * 12


def test_ctypes_mirrors_have_the_compiled_sizes(livo2):
    """every struct of the header, as compiled into the library, against its ctypes mirror (padding included)"""
    a = livo2.abi
    lib = a.load_library()
    mirrors = dict(livo2_state=a.State, livo2_map_view=a.MapView, livo2_lidar_cfg=a.LidarCfg, livo2_lidar_sums=a.LidarSums, livo2_lidar_points=a.LidarPoints,
                   livo2_lidar_result=a.LidarResult, livo2_cam=a.Cam, livo2_visual_cfg=a.VisualCfg, livo2_visual_sums=a.VisualSums, livo2_visual_step=a.VisualStep,
                   livo2_visual_result=a.VisualResult, livo2_plane_fit=a.PlaneFit, livo2_imu_cfg=a.ImuCfg, livo2_select_cfg=a.SelectCfg, livo2_retrieve_cfg=a.RetrieveCfg,
                   livo2_retrieve_candidates=a.RetrieveCandidates, livo2_retrieve_out=a.RetrieveOut, livo2_visual_obs=a.VisualObs,
                   livo2_retrieve_chain_out=a.RetrieveChainOut, livo2_frame_in=a.FrameIn, livo2_visual_reference=a.VisualReference)
    for name, cls in mirrors.items():
        assert lib.livo2_abi_sizeof(name.encode()) == C.sizeof(cls), name
    assert lib.livo2_abi_sizeof(b"livo2_imu_step") == 64 and lib.livo2_abi_sizeof(b"livo2_imu_pose") == 176
    assert lib.livo2_abi_sizeof(b"no_such_struct") == 0
    # every struct the header defines is covered by the query
    text = open(a.HEADER_PATH).read()
    for name in set(re.findall(r"typedef struct (livo2_[a-z0-9_]+)\s*\{", text)):       # (livo2_ctx is opaque)
        assert lib.livo2_abi_sizeof(name.encode()) > 0, name


def test_version_string(livo2):
    assert b"gfx950" in livo2.abi.load_library().livo2_version()


def test_no_device_means_error_not_fallback(livo2):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = livo2.abi.load_library().livo2_ctx_create(0, C.byref(h))
    assert rc == livo2.abi.ERR_NO_DEVICE and not h
    with pytest.raises(livo2.Livo2Error):
        livo2.Context(0)


def test_product_does_not_import_oracle():
    """the product package and the C sources must not reference oracle/ in any way"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "fast-livo2_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
    # nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg loads the oracle either
    for sub in ("tools", "scenarios", "include"):
        for dp, _, files in os.walk(os.path.join(root, sub)):
            for f in files:
                if f.endswith((".py", ".sh", ".h")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, os.path.join(sub, f)
                    assert "from tests" not in txt and "import tests" not in txt, os.path.join(sub, f)       # tests/helpers.py imports the oracle at module load
    import ast
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and n.module == "oracle" for n in ast.walk(fn))
        assert uses == (fn.name == "cpu_baseline"), fn.name
    assert not any(isinstance(n, ast.ImportFrom) and n.module == "oracle" for n in tree.body)
    # the timed GPU legs of bench.py take nothing from tests/ either: only the cpu_baseline legs do
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "tests" for n in ast.walk(fn))
        assert not uses or fn.name in ("cpu_baseline", "cpu_widened_rows"), fn.name
    assert not any(isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "tests" for n in tree.body)
    tree = ast.parse(open(os.path.join(root, "__graft_entry__.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and n.module == "oracle" for n in ast.walk(fn))
        assert uses == (fn.name == "smoke"), fn.name
