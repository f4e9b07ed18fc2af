"""Helpers for the test-only kernel emulator (tests/emu/): build + load libtdnet_emu.so through the same ctypes
binding the product uses, with numpy arrays standing in for HBM."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

_lib = None


def emu_lib():
    global _lib
    if _lib is None:
        import build_emu
        from tdnet_amd import _capi
        _lib = _capi.Lib(build_emu.build(), test_symbols=True)
    return _lib
