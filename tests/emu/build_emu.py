"""Build tests/emu/_build/libtdnet_emu.so: the product sources compiled for the HOST against the test-only
fiber emulator (tests/emu/td_device.h force-included ahead of the real header).  Test infrastructure only."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libtdnet_emu.so")
CXX = "/opt/rocm/lib/llvm/bin/clang++"
import glob  # noqa: E402
SRCS = glob.glob(os.path.join(ROOT, "tdnet_amd", "csrc", "*")) + \
       [os.path.join(HERE, f) for f in ("td_device.h", "tdemu.cpp", "build_emu.py")] + [os.path.join(ROOT, "include", "tdnet.h"), os.path.join(ROOT, "include", "tdnet_test.h")]


def build(force=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in SRCS):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # Host ISA extensions where the CPU has them: fmaf() / std::fma as one instruction instead of a libm call (the emulated fp32 MFMA is
    # two fmaf per output), _Float16 conversions in hardware.  Contraction stays off and both are exactly rounded: same results.
    try:
        flags = set(next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split())
    except (OSError, StopIteration):
        flags = set()
    isa = [f for f, need in (("-mfma", "fma"), ("-mavx2", "avx2"), ("-mf16c", "f16c")) if need in flags]
    cmd = [CXX, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unused-value", "-ffp-contract=off", "-Wno-psabi"] + isa + [
           "-include", os.path.join(HERE, "td_device.h"),
           "-x", "c++", os.path.join(ROOT, "tdnet_amd", "csrc", "td_model_test.hip"), os.path.join(HERE, "tdemu.cpp"), "-o", OUT]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
