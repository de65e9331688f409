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


def test_library_is_built_from_this_tree(lib_built):
    """iwae_build_id() = the sha256 csrc/build.sh stamps over the library's sources (16 hex digits): the binary under test is the one these
    sources build -- conftest.lib_built rebuilds on a mismatch -- and the id is readable from the file without loading it (bench.py quotes
    profile artefacts only when they carry the same id)."""
    from iwae_amd import _capi
    want = _capi.source_build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", want)
    assert _capi.file_build_id(lib_built) == want
    assert _capi.library_build_id() == want


def test_struct_layouts_match_header():
    from iwae_amd import _capi
    assert C.sizeof(_capi.Config) == 64          # uint32 + 7 int32 + uint64 + 6 int32; static_assert'ed in model.hip
    assert _capi.Config.struct_size.offset == 0
    assert _capi.Config.cond_dim.offset == 48
    assert _capi.Config.seed.offset == 32
    assert _capi.Config.precision.offset == 56
    assert _capi.Config().struct_size == 64      # the binding fills the ABI guard itself
    assert C.sizeof(_capi.Scalars) == 64
    assert C.sizeof(_capi.Tensors) == 12 * C.sizeof(C.c_void_p)


def _integration_stub():
    """The python code block of INTEGRATION.md section 3 ("Binding the C ABI directly")."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 3."):]
    return sec[sec.index("```python") + len("```python"):sec.index("```", sec.index("```python") + 10)]


def test_integration_md_stub_matches_the_abi(lib_built):
    """The hand-written ctypes stub a maintainer would copy out of INTEGRATION.md: its struct definitions are EXECUTED here
    and compared with the binding table (field by field) and with the sizes the header's static_asserts pin, and its
    iwae_create call must get past the struct_size guard (here, without a GPU, it then fails on the missing device)."""
    from iwae_amd import _capi
    src = _integration_stub()
    defs = src[:src.index("# ---- run")]
    ns = {"LIB_PATH": lib_built}
    exec(defs, ns)
    for name in ("Config", "Scalars"):
        stub, mine = ns[name], getattr(_capi, name)
        assert C.sizeof(stub) == C.sizeof(mine), name
        assert [(f[0], f[1]) for f in stub._fields_] == [(f[0], f[1]) for f in mine._fields_], name
    cfg = ns["make_config"](200, 100)
    assert cfg.struct_size == 64
    h = C.c_void_p()
    rc = ns["lib"].iwae_create(C.byref(cfg), C.byref(h))
    msg = ns["lib"].iwae_last_error().decode()
    import torch
    if torch.cuda.is_available():
        assert rc == 0, msg
        ns["lib"].iwae_destroy(h)
    else:
        assert rc == -2 and "struct_size" not in msg, (rc, msg)      # got past the ABI guard, stopped at "no HIP device"


def test_create_rejects_a_wrong_struct_size(lib_built):
    from iwae_amd import _capi
    lib = _capi.load()
    for bad in (0, 48, 56, 72):
        cfg = _capi.Config()
        cfg.n_layers, cfg.x_dim, cfg.world_size = 1, 784, 1
        cfg.n_hidden[0], cfg.n_latent[0] = 200, 100
        cfg.struct_size = bad
        h = C.c_void_p()
        assert lib.iwae_create(C.byref(cfg), C.byref(h)) == -1
        assert "struct_size" in lib.iwae_last_error().decode()
        assert not h.value


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
