"""Host-side defaults that the documentation promises (no GPU needed)."""
import importlib
import os


def test_needed_hint_cache_is_opt_in(monkeypatch):
    """The per-camera needed-segment hints pay only while the model stands still between two visits of a camera; a
    training loop revisits a camera once per epoch, where they cost time (HISTORY.md section 12): off unless asked for,
    and then for forwards that will not be differentiated (evaluation) only."""
    from vegs_amd import rasterizer
    monkeypatch.delenv("VEGS_RAST_HINTS", raising=False)
    r = importlib.reload(rasterizer)
    assert r._use_hints is False
    monkeypatch.setenv("VEGS_RAST_HINTS", "1")
    r = importlib.reload(rasterizer)
    assert r._use_hints == "eval"            # forwards under no_grad only: a training forward never gets a hint
    monkeypatch.setenv("VEGS_RAST_HINTS", "0")
    r = importlib.reload(rasterizer)
    assert r._use_hints is False
    assert r.needed_hints(True) is False and r.needed_hints("always") == "eval" and r.needed_hints(False) == "always"


def test_flag_constants_match_the_header():
    from vegs_amd import _capi
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    hdr = open(os.path.join(inc, "vegs_rast.h")).read() + open(os.path.join(inc, "vegs_rast_debug.h")).read()
    for name, value in (("VR_FLAG_DETERMINISTIC", _capi.FLAG_DETERMINISTIC), ("VR_FLAG_SCAN_BINNING", _capi.FLAG_SCAN_BINNING),
                        ("VR_FLAG_ROUNDS_OFF", _capi.FLAG_ROUNDS_OFF), ("VR_FLAG_ROUNDS_ON", _capi.FLAG_ROUNDS_ON)):
        shift = value.bit_length() - 1
        assert f"{name} = 1u << {shift}" in hdr, name
    assert f"#define VR_ABI_VERSION {_capi.ABI_VERSION}" in hdr
