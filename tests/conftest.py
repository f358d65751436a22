import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the native library and the oracle exist before any test imports them (building the
    checker / the product is not using a fallback: the product path itself still raises without its .so)."""
    import shutil
    lib = os.path.join(ROOT, "vegs_amd", "_lib", "libvegsrast.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        from vegs_amd import build
        build.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "libvr_oracle.so")) and shutil.which("gcc"):
        from oracle import oracle as orc
        orc.build()
