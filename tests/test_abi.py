"""The C-ABI library loads and exports every symbol include/gwen_hip.h declares (CPU only: no
compute call is made without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "gwen_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gwen_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("gwen_gcn_prep", "gwen_gcn_propagate_f32", "gwen_gcn_linear_f32", "gwen_hip_version"):
        assert must in names


def test_library_exports_every_declared_symbol(hip_lib):
    from gwen_amd import _lib
    names = declared_functions()
    for name in names:
        assert hasattr(hip_lib, name), f"libgwen_hip.so does not export {name}"
    # and the Python binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_strings(hip_lib):
    from gwen_amd import _lib
    assert _lib.version().startswith("gwen_hip ") and _lib.version().endswith("gfx950")
    assert hip_lib.gwen_hip_error_string(0) == b"success"
    assert b"invalid" in hip_lib.gwen_hip_error_string(-1)
    assert b"workspace" in hip_lib.gwen_hip_error_string(-3)


def test_argument_validation_without_gpu(hip_lib):
    # these return before any HIP call
    assert hip_lib.gwen_gcn_propagate_f32(None, None, None, None, None, None, -1, 4, 4, 4, 1, 0, 0, 0, None) == -1
    assert hip_lib.gwen_gcn_propagate_f32(None, None, None, None, None, None, 0, 4, 4, 4, 1, 0, 0, 0, None) == 0
    assert hip_lib.gwen_gcn_propagate_f32(None, None, None, None, None, None, 5, 4, 2, 4, 1, 0, 0, 0, None) == -1
    assert hip_lib.gwen_gcn_linear_f32(None, None, None, None, 5, 4, 4, 2, 4, 0, 0, None, 0, None) == -1
    assert hip_lib.gwen_gcn_linear_f32(None, None, None, None, 0, 4, 4, 4, 4, 0, 0, None, 0, None) == 0
    assert hip_lib.gwen_gcn_linear_workspace_floats(100000, 64, 64) == 0
    assert hip_lib.gwen_gcn_linear_workspace_floats(125, 16384, 1024) == 64 * 125 * 1024
    assert hip_lib.gwen_gcn_prep(None, None, -1, 0, 1, 1.0, 1, None, None, None, None, None, None, None, 0, None) == -1
    assert hip_lib.gwen_gcn_prep(8, None, 2 ** 31, 5, 1, 1.0, 1, 1, None, None, None, None, 1, None, 0, None) == -2
    assert hip_lib.gwen_relu_backward_f32(None, None, None, 0, None) == 0
    assert hip_lib.gwen_gcn_grad_workspace_floats(100, 8, 8) == 1
    assert hip_lib.gwen_gcn_grad_workspace_floats(5000, 8, 4) == 20 * 32 + 1      # 256-row chunks


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from gwen_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_code_object_is_gfx950_only(hip_lib):
    from gwen_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gwen_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libgcn_ref" not in text, f
