"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol that
include/fishtts_hip.h declares (no compute calls here — there is no GPU in this container)."""
import os
import re

import pytest


def _header_symbols(name="fishtts_hip.h"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ft_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from fish_tts_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    product, hooks = _header_symbols(), _header_symbols("fishtts_hip_test.h")
    assert product and hooks, "no symbols parsed from the headers"
    assert not set(product) & set(hooks)
    # the drop-in boundary carries no test / measurement hook
    assert not [n for n in product if n.startswith("ft_test_") or "profile" in n or n in ("ft_ar_set_noise", "ft_ar_get_debug")]
    declared = product + hooks
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(_lib.SYMBOLS), set(declared) ^ set(_lib.SYMBOLS)


def test_create_without_gpu_fails_loudly():
    import ctypes as C

    import torch

    from fish_tts_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    cfg = _lib.ft_ar_config()
    h = C.c_void_p()
    st = lib.ft_create(C.byref(cfg), None, 0, C.byref(h))
    assert st != _lib.FT_OK
    assert b"no CPU fallback" in lib.ft_last_error(None) or b"HIP" in lib.ft_last_error(None)


def test_struct_layout_matches_header():
    import ctypes as C

    from fish_tts_amd import _lib
    assert C.sizeof(_lib.ft_sampling) == 24
    assert C.sizeof(_lib.ft_ar_config) == 4 * 32
    assert _lib.ft_sampling.seed.offset == 16
