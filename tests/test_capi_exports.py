"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (the stable headers and
include/vegs_rast_debug.h, the experimental / test-only part), and the
product path fails loudly (no fallback) when asked to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    inc = os.path.join(ROOT, "include")
    src = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(vr_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_are_exported():
    from vegs_amd import _capi, build
    build.build()                                  # hipcc cross-compiles gfx950 without a GPU
    lib = ctypes.CDLL(_capi.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 8
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vegs_rast.h but not exported"
    assert set(declared) == set(_capi.EXPORTS), (declared, _capi.EXPORTS)
    assert _capi.load().vr_abi_version() == _capi.ABI_VERSION


def test_struct_layouts_match_header_sizes():
    """ctypes mirrors of the ABI structs: pointer-sized fields and 4-byte scalars, no surprises."""
    from vegs_amd import _capi
    p = ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_capi.VrSettings) == 8 * 4 + 4 * p + 8          # + uint32 flags, padded to pointer alignment
    assert ctypes.sizeof(_capi.VrInputs) == 2 * 4 + 9 * p + 8      # + shs_tail, tail_start (ABI v6)
    assert ctypes.sizeof(_capi.VrOutputs) == 6 * p
    assert ctypes.sizeof(_capi.VrSaved) == 4 * p + 3 * 8 + 8        # + uint64 ticket (ABI v5)
    assert ctypes.sizeof(_capi.VrOutGrads) == 5 * p
    assert ctypes.sizeof(_capi.VrInGrads) == 11 * p               # + dL_dshs_tail (ABI v6)
    assert ctypes.sizeof(_capi.VrCounters) == 5 * 8


def test_flag_values_agree_between_header_binding_and_kernels():
    """VrFlags is defined three times -- include/vegs_rast.h (the contract), vegs_amd/_capi.py (the ctypes binding) and
    csrc/vr_device.h (what the kernels test; tied to the header by a static_assert in api.hip) -- and a renumbering in one of
    them would silently flip e.g. the default tile-list semantics.  The header's values are parsed and compared with both."""
    from vegs_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "vegs_rast.h")).read() + open(os.path.join(ROOT, "include", "vegs_rast_debug.h")).read()
    header = {m.group(1): 1 << int(m.group(2)) for m in re.finditer(r"\bVR_(FLAG_[A-Z_]+)\s*=\s*1u\s*<<\s*(\d+)", hdr)}
    assert len(header) >= 12 and len(set(header.values())) == len(header)          # distinct bits
    dev = open(os.path.join(ROOT, "vegs_amd", "csrc", "vr_device.h")).read()
    device = {m.group(1): 1 << int(m.group(2)) for m in re.finditer(r"constexpr uint32_t (FLAG_[A-Z_]+)\s*=\s*1u\s*<<\s*(\d+)", dev)}
    api = open(os.path.join(ROOT, "vegs_amd", "csrc", "api.hip")).read()
    for name, value in header.items():
        assert getattr(_capi, name) == value, name
        assert device.get(name) == value, name
        assert f"{name} == VR_{name}" in api, f"{name} is missing from api.hip's static_assert"
        assert re.search(rf"KNOWN_FLAGS\s*=[^;]*\b{name}\b", api, re.S), f"{name} is missing from KNOWN_FLAGS"
    bits = [device[k] for k in re.findall(r"constexpr uint32_t (FLAG_[A-Z_]+)", dev)]
    assert bits == sorted(bits)                                                      # listed in bit order


def test_invalid_arguments_are_reported_without_a_gpu():
    from vegs_amd import _capi
    lib = _capi.load()
    rc = lib.vr_forward(None, None, None, _capi.VrAllocFn(lambda u, k, n: 0), None, None, None)
    assert rc == -1 and b"required" in lib.vr_last_error()
    rc = lib.vr_mark_visible(None, 5, None, None, None, None)
    assert rc == -1


def test_product_path_has_no_cpu_fallback():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False)
    r = GaussianRasterizer(raster_settings=rs)
    with pytest.raises(Exception, match="GPU"):
        r(means3D=torch.zeros(2, 3), means2D=torch.zeros(2, 3), opacities=torch.ones(2, 1), shs=torch.zeros(2, 16, 3),
          scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    with pytest.raises(Exception, match="GPU"):
        r.markVisible(torch.zeros(2, 3))
    with pytest.raises(Exception, match="exactly one"):
        r(means3D=torch.zeros(2, 3), means2D=torch.zeros(2, 3), opacities=torch.ones(2, 1))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under vegs_amd/ or diff_gaussian_rasterization/ may use it."""
    for pkg in ("vegs_amd", "diff_gaussian_rasterization", "simple_knn", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".c")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in txt and "from oracle" not in txt and "vr_oracle" not in txt, f


def test_arena_is_freed_by_refcount_not_by_gc():
    """The allocator arena must not sit in a reference cycle: a cycle kept ~0.5 GB of per-call buffers
    alive until the cyclic GC ran, which sent the caching allocator back to hipMalloc every other step."""
    import gc
    import weakref
    from vegs_amd import _capi
    gc.disable()
    try:
        arena = _capi.Arena(torch.device("cpu"))
        cb = arena.callback()
        assert cb(None, _capi.VR_BUF_SCRATCH, 64) != 0 and cb(None, _capi.VR_BUF_GEOM, 64) != 0
        ref = weakref.ref(arena)
        del cb
        kept = arena.kept
        del arena
        assert ref() is None, "Arena is part of a reference cycle"
        assert _capi.VR_BUF_GEOM in kept
    finally:
        gc.enable()
