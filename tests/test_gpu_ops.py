"""GPU parity, operator level: each hand-written HIP kernel family through the C ABI vs the same op in PyTorch fp32
on CPU, on ragged small shapes and on the real TDNet shapes."""
import pytest
import torch

import opcheck
from tdnet_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    return _capi.test_lib()                      # the in-tree libtdnet_hip.so; raises if it is missing


@pytest.fixture(scope="module")
def mem():
    return opcheck.TorchMem()


DIRECT = {"winograd": 0}     # force the direct implicit-GEMM kernels (the library default routes wide stride-1 3x3 convs to Winograd)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_conv_variants(lib, mem, tile):
    opcheck.conv(lib, mem, 13, 21, 64, 128, 3, 1, 2, 1, True, tile, opts=DIRECT)
    opcheck.conv(lib, mem, 13, 21, 32, 96, 3, 2, 1, 0, False, tile, opts=DIRECT)
    opcheck.conv(lib, mem, 11, 19, 64, 19, 1, 1, 1, 2, False, tile, opts=DIRECT)
    opcheck.conv(lib, mem, 17, 9, 128, 64, 1, 2, 1, 0, True, tile, opts=DIRECT)
    opcheck.conv(lib, mem, 12, 30, 64, 160, 3, 1, 4, 1, False, tile, opts=DIRECT)
    opcheck.conv(lib, mem, 97, 193, 64, 128, 3, 1, 2, 1, True, tile, opts=DIRECT)      # native 769x1537 feature size, many ragged tiles
    opcheck.conv(lib, mem, 7, 9, 32, 64, 1, 1, 1, 0, False, tile, opts=DIRECT)         # 1, 2, 3 K steps: pipeline prologue / odd tail
    opcheck.conv(lib, mem, 7, 9, 64, 64, 1, 1, 1, 0, True, tile, opts=DIRECT)
    opcheck.conv(lib, mem, 7, 9, 96, 64, 1, 1, 1, 1, False, tile, opts=DIRECT)
    opcheck.conv(lib, mem, 128, 256, 512, 512, 3, 1, 4, 1, True, tile, tol=2e-4, opts=DIRECT)   # the dominant layer4 shape on every variant


@pytest.mark.parametrize("wino", [0, 1, 3])
def test_conv_real_shapes(lib, mem, wino):
    o = {"winograd": wino}
    k = 3.0 if wino == 3 else 1.0                                         # F(4x4,3x3): per-conv rounding error ~6x F(2x2)'s
    opcheck.conv(lib, mem, 64, 128, 64, 64, 3, 1, 1, 1, True, opts=o)             # layer1-like
    opcheck.conv(lib, mem, 64, 128, 64, 128, 3, 2, 1, 1, False, opts=o)           # layer2.0.conv1
    opcheck.conv(lib, mem, 64, 128, 64, 128, 1, 2, 1, 0, False, opts=o)           # layer2.0.downsample
    opcheck.conv(lib, mem, 64, 128, 128, 128, 3, 1, 1, 1, True, tol=k * 1e-4, opts=o)    # layer2 (Winograd from mode 3 on)
    opcheck.conv(lib, mem, 32, 64, 256, 256, 3, 1, 2, 1, True, tol=k * 1e-4, opts=o)     # layer3
    opcheck.conv(lib, mem, 32, 64, 256, 512, 3, 1, 4, 1, False, tol=k * 1e-4, opts=o)    # layer4.0.conv1
    opcheck.conv(lib, mem, 32, 64, 512, 512, 3, 1, 8, 1, True, tol=k * 2e-4, opts=o)     # layer4.1.conv1, K = 4608
    opcheck.conv(lib, mem, 32, 64, 512, 512, 3, 1, 16, 1, True, tol=k * 2e-4, opts=o)    # resnet34 multi-grid 16
    opcheck.conv(lib, mem, 128, 256, 512, 512, 3, 1, 4, 1, True, tol=k * 2e-4, opts=o)   # the dominant kernel at its real size
    opcheck.conv(lib, mem, 128, 256, 512, 64, 1, 4, 1, 2, False, opts=o)          # w_ks.0 on the stride-4 key grid
    opcheck.conv(lib, mem, 1, 2048, 512, 512, 1, 1, 1, 0, False, opts=o)          # attention fc on the cached value matrix
    opcheck.conv(lib, mem, 40, 40, 512, 128, 3, 1, 1, 1, False, tol=k * 1e-4, opts=o)    # FCNHead conv


def test_stem(lib, mem):
    opcheck.stem(lib, mem, 33, 65)
    opcheck.stem(lib, mem, 257, 513)
    opcheck.stem(lib, mem, 300, 422)
    for (H, W) in ((257, 513), (300, 422), (1024, 2048), (34, 66)):
        opcheck.stem(lib, mem, H, W, opts={"fusion": 16})


def test_attention(lib, mem):
    opcheck.attention(lib, mem, 45, 6, 512)
    opcheck.attention(lib, mem, 300, 200, 512, spike=True)
    opcheck.attention(lib, mem, 200, 131, 128, True, False, qk_scale=2.0)
    opcheck.attention(lib, mem, 1, 1, 128)
    opcheck.attention(lib, mem, 2048, 2048, 512)                           # cached-frame propagation step
    opcheck.attention(lib, mem, 18721, 1225, 512)                          # native 769x1537: ragged Lq and Lk
    opcheck.attention(lib, mem, 8192, 512, 128, qk_scale=1.5)              # td2 @512x1024
    opcheck.attention(lib, mem, 32768, 2048, 512, spike=True)              # final step @1024x2048
    opcheck.attention(lib, mem, 32768, 2048, 128, qk_scale=1.5)            # td2-psp18 @1024x2048 (BASELINE configs[1]): the d_v = 128 variant at full size
    opcheck.attention(lib, mem, 18721, 1225, 128, spike=True)              # td2 at the native 769x1537


def test_attention_online_and_layernorm_statistics(lib, mem):
    for online in (1, 2, 0):
        opcheck.attention(lib, mem, 300, 200, 512, spike=True, online=online, ln=True)
        opcheck.attention(lib, mem, 97, 300, 512, ramp=True, online=online, ln=True)
        opcheck.attention(lib, mem, 18721, 1225, 512, online=online, ln=True)
        opcheck.attention(lib, mem, 32768, 2048, 512, spike=True, online=online, ln=True)
        opcheck.attention(lib, mem, 32768, 2048, 128, qk_scale=1.5, online=online, ln=True)
        opcheck.attention(lib, mem, 18721, 1225, 128, spike=True, ramp=True, online=online, ln=True)


def test_layernorm_ppm_upsample(lib, mem):
    for hw, c in [(45, 512), (153, 128), (1000, 512), (32768, 512), (18721, 128)]:
        opcheck.layernorm(lib, mem, hw, c)
    for h, w, pid in [(5, 9, 0), (9, 17, 1), (13, 25, 1), (97, 193, 0), (128, 256, 1)]:
        opcheck.ppm(lib, mem, h, w, pid)
    for c, h, w, H, W in [(19, 5, 9, 33, 65), (19, 97, 193, 769, 1537), (19, 128, 256, 1024, 2048)]:
        opcheck.upsample(lib, mem, c, h, w, H, W)
