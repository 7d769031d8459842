"""The C-ABI library loads and exports every symbol include/legged_hip.h declares; ctypes struct layouts
match the C ones.  No compute calls (no GPU needed)."""
import ctypes
import os
import re

import pytest

from legged_games_gym_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))


def _declared_functions():
    text = open(os.path.join(REPO, "include", "legged_hip.h")).read()
    return sorted(set(re.findall(r"\b(lg_[a-z_]+)\s*\(", text)) - {"lg_sim"})


def test_header_symbol_list_matches_binding():
    assert _declared_functions() == sorted(capi.EXPORTED_SYMBOLS)


def test_hip_library_exports_all_symbols_and_layouts():
    path = capi.library_path()
    if not os.path.isfile(path):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(path)
    for sym in _declared_functions():
        assert hasattr(lib, sym), sym
    capi.bind_prototypes(lib, "lg_")          # raises on any struct size mismatch
    assert lib.lg_abi_version() == capi.LG_ABI_VERSION


def test_oracle_exports_same_abi(oracle_lib):
    for sym in _declared_functions():
        if sym.startswith(("lg_policy_", "lg_mlp_")) or sym in ("lg_step_policy", "lg_rollout_policy", "lg_gae_returns", "lg_ppo_loss", "lg_ppo_minibatch", "lg_adam_step", "lg_rollout_record", "lg_rollout_finish"):
            continue          # learner-side kernels: their reference is torch fp32 (forward / autograd), not the C oracle
        if sym in ("lg_set_deferred_extras", "lg_extras_flush", "lg_device_status", "lg_clear_device_status", "lg_debug_handover", "lg_resample_reset_commands"):
            continue          # launch scheduling / wave hand-over status of the device library: nothing to restate on the CPU
        assert hasattr(oracle_lib, "lgo_" + sym[3:]), sym
    assert oracle_lib.lgo_abi_version() == capi.LG_ABI_VERSION


def test_product_never_touches_the_oracle():
    """The product path must not import / load anything under oracle/ (parity claims depend on it)."""
    pkg = os.path.join(REPO, "legged_games_gym_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(root, f)
                assert "liblg_oracle" not in text and "lgo_" not in text.replace('"lgo_"', ""), os.path.join(root, f)


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "library_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load_library()


def test_cpu_device_is_refused():
    import torch
    from legged_games_gym_amd.device_sim import DeviceSim
    with pytest.raises(RuntimeError, match="no CPU product path"):
        DeviceSim(None, None, None, torch.device("cpu"))
