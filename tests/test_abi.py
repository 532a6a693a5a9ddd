"""The C-ABI library builds for gfx950, loads, and exports every symbol include/vl3d.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from videoloop3d_amd import _lib
    return _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "vl3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vl3d_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built):
    syms = header_symbols()
    assert len(syms) >= 15
    l = ctypes.CDLL(built.LIB_PATH)
    for s in syms:
        assert hasattr(l, s), f"{s} declared in include/vl3d.h but not exported"


def test_binding_covers_header(built):
    assert sorted(built.SIGNATURES) == header_symbols()


def test_no_gpu_calls_needed_for_metadata(built):
    assert built.lib().vl3d_version() >= 100


def test_struct_layout_matches_header(built):
    # 25 x 4-byte fields / (10 x 4 + pad + 6 x 8 + 4 + pad): catches accidental drift between vl3d.h and ctypes
    assert ctypes.sizeof(built.RenderDesc) == 104
    assert ctypes.sizeof(built.LossDesc) == 96
    assert ctypes.sizeof(built.Stage1ObjectiveDesc) == 64      # 4 x int32 + 12 x float


def test_cpu_tensor_is_rejected_loudly(built):
    import torch
    from videoloop3d_amd.render import render_planes
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        render_planes(torch.zeros(2, 1, 8, 8, 4), torch.eye(3).expand(2, 3, 3), 8, 8)


def test_a_library_built_from_other_sources_is_refused(tmp_path):
    """__graft_entry__.build() stamps the library with the sha256 of its sources; _lib refuses a library whose stamp names other sources (a stale
    prebuilt .so beside edited kernels would otherwise answer with old code behind unchanged symbols) and build() recompiles instead of trusting it."""
    import pytest
    import __graft_entry__ as ge
    from videoloop3d_amd import _lib as L
    ge.build()
    assert open(ge.STAMP).read().strip() == ge.source_hash() == L.sources_sha256()
    L.check_stamp(ge.LIB)                                           # the tree's own library passes
    (tmp_path / "libvl3d_hip.stamp").write_text("0" * 64 + "\n")
    with pytest.raises(RuntimeError, match="other sources"):
        L.check_stamp(str(tmp_path / "libvl3d_hip.so"))
