"""CPU tests of the drop-in boundary: the in-tree library loads, exports every symbol
include/iwae_amd.h declares, the ctypes structs match the header, and -- with no GPU in this
container -- model creation fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "iwae_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(iwae_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = _header_functions()
    for must in ("iwae_create", "iwae_destroy", "iwae_forward", "iwae_train_step", "iwae_forward_backward", "iwae_adam_step",
                 "iwae_eval_llh", "iwae_decode", "iwae_set_params", "iwae_get_params", "iwae_grad_devptr", "iwae_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_built):
    lib = C.CDLL(lib_built)
    for name in _header_functions():
        assert hasattr(lib, name), "libiwae_amd.so does not export %s" % name


def test_binding_table_covers_header(lib_built):
    from iwae_amd import _capi
    assert sorted(_capi.SYMBOLS) == _header_functions()
    _capi.load()


def test_struct_layouts_match_header():
    from iwae_amd import _capi
    assert C.sizeof(_capi.Config) == 56          # 7 int32 (+4 pad) + uint64 + 4 int32; static_assert'ed in model.hip
    assert _capi.Config.cond_dim.offset == 48
    assert _capi.Config.seed.offset == 32
    assert C.sizeof(_capi.Scalars) == 64
    assert C.sizeof(_capi.Tensors) == 12 * C.sizeof(C.c_void_p)


def test_no_cpu_fallback_without_gpu(lib_built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from iwae_amd.native import NativeModel
    with pytest.raises(Exception) as ei:
        NativeModel(1, 200, 100)
    assert "HIP" in str(ei.value) or "GPU" in str(ei.value) or "hip" in str(ei.value)


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "iwae_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), "%s references the oracle" % fn
    assert "oracle" not in open(os.path.join(ROOT, "main.py")).read()
