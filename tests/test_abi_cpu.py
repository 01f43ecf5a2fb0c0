"""CPU: the C-ABI library loads and exports every symbol include/flair_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "flair_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flair_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    names = declared_symbols()
    for must in ("flair_conv_nhwc", "flair_groupnorm_nhwc", "flair_qkv_attention", "flair_temporal_attention",
                 "flair_dcn_align", "flair_flow_warp", "flair_timestep_embedding", "flair_sampler_update",
                 "flair_depthwise_filter", "flair_jpeg_roundtrip", "flair_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from flair_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.flair_abi_version() >= 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from flair_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.FlairHipUnavailable, match="no fallback"):
        _lib.lib()


def test_argument_errors_are_reported_without_a_gpu():
    """Validation happens before any launch, so error plumbing is testable on CPU."""
    from flair_amd import _lib
    lib = _lib.lib()
    p = _lib.ConvParams()
    p.dtype = 7
    rc = lib.flair_conv_nhwc(ctypes.byref(p), (ctypes.c_void_p * 4)(1, 0, 0, 0), ctypes.c_void_p(16), None, None,
                             None, None, ctypes.c_void_p(16), None, ctypes.c_size_t(0), None)
    assert rc == -1 and b"bad dtype" in lib.flair_last_error()


def test_bcast_entry_validates_arguments():
    """flair_bcast_weights (the path's only collective) refuses a null blob / communicator with a message and
    does not touch RCCL for that (no GPU and no librccl needed here)."""
    import ctypes
    from flair_amd import _lib
    lib = _lib.lib()
    rc = lib.flair_bcast_weights(None, ctypes.c_size_t(0), 0, None, None)
    assert rc == -1 and b"flair_bcast_weights" in lib.flair_last_error()
