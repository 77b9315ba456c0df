"""CPU: the C-ABI library loads and exports exactly what include/goalnet_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from cvml_goalnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "goalnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(goalnet_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.PROTOTYPES.keys())


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} missing from libgoalnet_hip.so"
    loaded = _lib.load()
    assert loaded.goalnet_abi_version() == _lib.ABI_VERSION
    assert loaded.goalnet_last_error() is not None


def test_argument_errors_do_not_need_a_gpu():
    """Shape/NULL checks run before any launch and return negative codes with a message."""
    lib = _lib.load()
    rc = lib.goalnet_conv3x3_fwd(None, None, None, None, None, 0, None, 1, 8, 8, 64, 64, None, 0, None, 0, None)
    assert rc == -1 and b"null" in lib.goalnet_last_error()
    # ten 11x11 frames (the reference's sub-batch at 40x40) need split-K slabs, 1024 frames of 72x72 do not
    assert lib.goalnet_conv3x3_fwd_ws_bytes(10, 11, 11, 256, 512) > 0
    assert lib.goalnet_conv3x3_fwd_ws_bytes(1024, 72, 72, 256, 512) == 0
    assert lib.goalnet_conv3x3_fwd_bf16p_ws_bytes(1024, 72, 72, 256, 512) == 0
    rc = lib.goalnet_adam_step_dev(16, 16, 16, 16, 4, 1e-3, 0.9, 0.999, 1e-8, None, 1, 1.0, None)
    assert rc == -1 and b"step counter" in lib.goalnet_last_error()
    rc = lib.goalnet_rows_gather(16, 16, 6, 1, 16, None)
    assert rc == -2 and b"multiple of 4" in lib.goalnet_last_error()
    rc = lib.goalnet_linear_fwd(16, 64, None, None, 0, 16, None, 0, None, 0, 16, 64, None, 0, 4, 33, 64, None, 0, None)
    assert rc == -2 and b"multiple" in lib.goalnet_last_error()
    assert lib.goalnet_linear_fwd_ws_bytes(1024, 2508800, 512) > 0
    assert lib.goalnet_conv3x3_wgrad_ws_bytes(8, 72, 72, 256, 512) > 0


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cvml_goalnet_amd import AVM, GoalnetError
    m = AVM(audio_included=True)
    with pytest.raises(GoalnetError):
        m(torch.zeros(2, 30, 30), torch.zeros(2, 3, 40, 40))
