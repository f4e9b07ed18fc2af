#!/usr/bin/env python3
"""CPU-only: the kernel SOURCES under AddressSanitizer, through the test-only fiber emulator (tests/emu/).  GPU ASan is not available on this pool; here every
"global" buffer is a malloc'ed / numpy array with redzones and a workgroup's LDS is a heap block, so an out-of-range raw pointer access, an LDS overrun or a
use-after-free in the host code shows up.  (Range-checked buffer_load / buffer_store go through the emulator's bounds check, as on the hardware.)

    python tools/emu_asan.py build            # tests/emu/_build/libtdnet_emu_asan.so (clang++ -O1 -g -fsanitize=address, ~2 min)
    python tools/emu_asan.py run [what ...]   # what: adb3 attn gemm fp32 pipeline (default: all); re-executes itself with the ASan runtime preloaded
"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
OUT = os.path.join(EMU, "_build", "libtdnet_emu_asan.so")
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [CXX, "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unused-value", "-ffp-contract=off",
           "-Wno-psabi", "-mfma", "-mavx2", "-mf16c", "-include", os.path.join(EMU, "td_device.h"), "-x", "c++",
           os.path.join(ROOT, "tdnet_amd", "csrc", "td_model_test.hip"), os.path.join(EMU, "tdemu.cpp"), "-o", OUT]
    subprocess.run(cmd, check=True)
    print(OUT)


def run(what):
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import numpy as np
    import opcheck
    from tdnet_amd import _capi, arch, weights
    from tdnet_amd.engine import Engine
    lib = _capi.Lib(OUT, test_symbols=True)
    MEM = opcheck.NumpyMem()
    o2, o3 = {"precision": 2}, {"precision": 3}
    for w in what:
        if w == "adb3":                                          # td_conv_ad_b3.h: narrow convs + the packed-row stem, ragged rows / channels, two column tiles
            opcheck.conv(lib, MEM, 13, 21, 64, 64, 3, 1, 1, 1, True, opts=o2)
            opcheck.conv(lib, MEM, 11, 9, 96, 48, 3, 2, 2, 0, False, opts=o2)
            opcheck.conv(lib, MEM, 9, 17, 64, 100, 1, 2, 1, 0, True, opts=o2)
            opcheck.conv(lib, MEM, 13, 21, 64, 128, 3, 2, 1, 1, False, opts=dict(o2, winograd=0))
            for H, W in ((33, 65), (7, 9)):
                opcheck.stem(lib, MEM, H, W, opts=o2)
        elif w == "attn":                                        # td_attn_b3.h: both forms, ragged query / key tiles, LayerNorm strips past Lq
            opcheck.attention(lib, MEM, 45, 6, 512, online=18)
            opcheck.attention(lib, MEM, 97, 130, 512, True, True, spike=True, online=18, ln=True)
            opcheck.attention(lib, MEM, 33, 1, 512, online=18, ln=True)
            opcheck.attention(lib, MEM, 70, 300, 512, online=17, ramp=True, ln=True)
            opcheck.attention(lib, MEM, 130, 193, 128, True, True, spike=True, online=17, ln=True)
        elif w == "gemm":                                        # td_gemm_b3.h: Winograd GEMMs whole and in row-parity chunks, ragged N, 1x1 roles
            opcheck.conv(lib, MEM, 13, 21, 128, 128, 3, 1, 2, 1, True, opts=o3)
            opcheck.conv(lib, MEM, 20, 30, 256, 132, 3, 1, 4, 1, True, opts=dict(o3, overlap=41 | 4))
            opcheck.conv(lib, MEM, 11, 19, 64, 160, 1, 1, 1, 2, True, opts=o3)
            opcheck.conv(lib, MEM, 7, 9, 64, 128, 1, 1, 1, 0, True, opts=o3)
        elif w == "fp32":                                        # the default kernels at a glance: direct, Winograd (+ chunks), LDS-DMA GEMM, attention, tail
            opcheck.conv(lib, MEM, 13, 21, 64, 128, 3, 1, 2, 1, True, opts={"winograd": 0})
            opcheck.conv(lib, MEM, 20, 30, 256, 132, 3, 1, 4, 1, True, opts={"overlap": 41 | 4})
            opcheck.conv(lib, MEM, 13, 21, 64, 64, 3, 1, 1, 1, True)
            opcheck.stem(lib, MEM, 33, 65)
            opcheck.attention(lib, MEM, 97, 130, 512, True, True, spike=True, online=2, ln=True)
            opcheck.ppm(lib, MEM, 9, 17, 1)
            opcheck.upsample(lib, MEM, 19, 5, 9, 33, 65)
        elif w == "pipeline":                                    # whole frames: warm-up + steady state, default and precision 3 with the chains forced on
            H, W = 33, 65
            spec = arch.model_spec("td4", 19, "resnet18")
            sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
            for opts in ({"overlap": 41 | 4}, {"precision": 3, "overlap": 41 | 4}):
                e = Engine(4, 18, 19, H, W, 0, lib=lib, opts=opts)
                e.load_state_dict(sd)
                for t, x in enumerate(weights.synth_video(H, W, 6, seed=2)):
                    out = np.zeros((1, 19, H, W), np.float32)
                    e.forward(x, t % 4, out)
                e.close()
        else:
            raise SystemExit("unknown: " + w)
        print(w, "ok", flush=True)


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        what = sys.argv[2:] or ["adb3", "attn", "gemm", "fp32", "pipeline"]
        rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")[0]
        if os.environ.get("LD_PRELOAD", "") != rt:
            env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0")
            sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), "run"] + what, env=env).returncode)
        run(what)
