"""The product library loads and exports every symbol include/tdnet.h declares; strict weight loading and argument
validation behave like the reference's constructor / load_state_dict(strict=True).  No GPU compute here."""
import ctypes
import os
import re

import numpy as np
import pytest

import emu_util
from tdnet_amd import _capi, arch, weights
from tdnet_amd.engine import Engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_library_exports_every_declared_symbol():
    """The PRODUCT library exports exactly what include/tdnet.h declares -- and none of the tests' single-operator entry points or probes
    (include/tdnet_test.h: tdnet_op_*, tdnet_bench_*), which live in libtdnet_hip_test.so, a superset built from the same sources."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    lib = _capi.Lib(_capi.DEFAULT_LIB)
    hdr = open(os.path.join(ROOT, "include", "tdnet.h")).read()
    declared = set(re.findall(r"\b(tdnet_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib.dll, name), name
    assert b"gfx950" in lib.tdnet_version()
    thdr = open(os.path.join(ROOT, "include", "tdnet_test.h")).read()
    tdeclared = set(re.findall(r"\b(tdnet_[a-z0-9_]+)\s*\(", thdr))
    assert tdeclared == set(_capi.TEST_SYMBOLS) and not (tdeclared & declared), tdeclared ^ set(_capi.TEST_SYMBOLS)
    exported = set(re.findall(r" T (tdnet_[a-z0-9_]+)", subprocess.run(["nm", "-D", "--defined-only", _capi.DEFAULT_LIB], stdout=subprocess.PIPE, check=True).stdout.decode()))
    assert exported == declared, exported ^ declared                     # nothing but the boundary: no tdnet_op_*, no tdnet_bench_*
    tlib = _capi.Lib(_capi.TEST_LIB, test_symbols=True)
    for name in declared | tdeclared:
        assert hasattr(tlib.dll, name), name
    from tdnet_amd import build as b
    assert b.built_hash(_capi.TEST_LIB) == b.built_hash(_capi.DEFAULT_LIB) == b.source_hash()


def test_library_is_stamped_with_the_hash_of_its_sources(tmp_path):
    """tdnet_amd/build.py: the .so carries a hash of csrc/* + include/tdnet.h (compiled into tdnet_version()); build() rebuilds when the
    stamp and the sources disagree instead of trusting mtimes, and the stamp is readable without loading the library."""
    import __graft_entry__ as g
    from tdnet_amd import build as b
    g.build()
    want = b.source_hash()
    assert b.built_hash() == want
    assert _capi.Lib(_capi.DEFAULT_LIB).tdnet_version().decode().endswith("tdnet-src-hash:" + want)
    # a library built from other sources is recognised as stale (a copy with one stamp byte changed stands in for it)
    blob = open(b.OUT, "rb").read()
    i = blob.index(b.STAMP_MARK) + len(b.STAMP_MARK)
    fake = tmp_path / "stale.so"
    fake.write_bytes(blob[:i] + (b"0" if blob[i:i + 1] != b"0" else b"1") + blob[i + 1:])
    assert b.built_hash(str(fake)) not in (None, want)


def test_product_fails_loudly_without_library(tmp_path):
    with pytest.raises(_capi.TdnetError):
        _capi.Lib(str(tmp_path / "missing.so"))


def test_create_rejects_bad_configs():
    lib = emu_util.emu_lib()
    for cfg in [(3, 18, 19, 65, 65), (4, 101, 19, 65, 65), (2, 101, 19, 65, 65), (1, 18, 19, 65, 65), (4, 18, 0, 65, 65), (4, 18, 19, 4, 65)]:
        with pytest.raises(_capi.TdnetError):
            Engine(*cfg, 0, lib=lib)


def test_strict_state_dict_loading():
    """host logic of tdnet_set_weight / tdnet_finalize_weights (td4_psp18.py:236-237 strict=True semantics)."""
    lib = emu_util.emu_lib()
    spec = arch.model_spec("td2", 19, "resnet18")
    H, W = 33, 65
    h, w = arch.feat_size(H), arch.feat_size(W)
    sd = weights.synth_state_dict(spec, h, w, 0)
    e = Engine(2, 18, 19, H, W, 0, lib=lib)
    with pytest.raises(_capi.TdnetError, match="Unexpected key"):
        e.load_state_dict({"bogus.weight": np.zeros(3, np.float32)})
    e = Engine(2, 18, 19, H, W, 0, lib=lib)
    bad = dict(sd)
    bad["head1.conv5.4.weight"] = np.zeros((19, 63, 1, 1), np.float32)
    with pytest.raises(_capi.TdnetError, match="size mismatch"):
        e.load_state_dict(bad)
    e = Engine(2, 18, 19, H, W, 0, lib=lib)
    missing = dict(sd)
    del missing["atn2.fc.0.conv.bias"]
    with pytest.raises(_capi.TdnetError, match="Missing key"):
        e.load_state_dict(missing)
    # forward before weights are finalized must fail, not run on garbage
    e = Engine(2, 18, 19, H, W, 0, lib=lib)
    with pytest.raises(_capi.TdnetError, match="not finalized"):
        e.forward(np.zeros((1, 3, H, W), np.float32), 0, np.zeros((1, 19, H, W), np.float32))
    # LayerNorm affine of the wrong plane size = the reference's LayerNorm([97,193]) failure at other resolutions
    e = Engine(2, 18, 19, 65, 129, 0, lib=lib)
    with pytest.raises(_capi.TdnetError, match="size mismatch"):
        e.load_state_dict(sd)


def test_flops_accounting_matches_survey():
    """Algorithmic FLOP per steady-state frame, computed by the library's own accounting of the reference's op list, against the
    figures SURVEY.md 8d measured by hook-counting the reference itself: C3 td4-psp18 1024x2048 936.2 GFLOP, C2 td2-psp18 1024x2048
    809.1, native td4-psp18 769x1537 514.9, C1 td2-psp18 512x1024 197.5, td2-psp34 720x960 484.0, td2-psp50 769x1537 1103.0."""
    lib = emu_util.emu_lib()
    for (m, bb, H, W, expect) in [(4, 18, 1024, 2048, 936.2), (2, 18, 1024, 2048, 809.1), (4, 18, 769, 1537, 514.9),
                                  (2, 18, 512, 1024, 197.5), (2, 34, 720, 960, 484.0), (2, 50, 769, 1537, 1103.0)]:
        spec = arch.model_spec("td%d" % m, 19, "resnet%d" % bb)
        e = Engine(m, bb, 19, H, W, 0, lib=lib, opts={"winograd": 0})          # the count is of the reference's ops, not of the kernels'
        e.load_state_dict(weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0))
        got = e.flops_per_frame() / 1e9
        assert abs(got - expect) <= 1e-3 * expect, (m, bb, H, W, got, expect)
        e.close()


def test_model_classes_mirror_reference_api():
    import torch
    from tdnet_amd.model import td4_psp18, td2_psp50
    with pytest.raises(AssertionError):
        td4_psp18.td4_psp18(nclass=19, path_num=2, synthetic_seed=0)
    with pytest.raises(AssertionError):
        td2_psp50.td2_psp50(nclass=19, path_num=2, backbone="vgg", synthetic_seed=0)
    with pytest.raises(FileNotFoundError):
        td4_psp18.td4_psp18(nclass=19, path_num=4, model_path="/nonexistent/td4-psp18.pkl")
    m = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, synthetic_seed=0).eval()
    with pytest.raises(_capi.TdnetError):                        # no CPU fallback
        m(torch.zeros(1, 3, 33, 65), pos_id=0)
    with pytest.raises(_capi.TdnetError):
        m.forward_labels(torch.zeros(1, 3, 33, 65), pos_id=0)
    with pytest.raises(_capi.TdnetError):                        # a batch is N streams (td4_psp18.py:216-229) -- and still needs the GPU
        m.forward_labels(torch.zeros(2, 3, 33, 65), pos_id=0)
    with pytest.raises(RuntimeError):                            # forward_labels validates like forward: batch, pos_id, tensor
        m.forward_labels(torch.zeros(0, 3, 33, 65), pos_id=0)
    with pytest.raises(RuntimeError):
        m.forward_labels(torch.zeros(1, 3, 33, 65), pos_id=7)
    with pytest.raises(RuntimeError):
        m.forward_labels([[1.0]], pos_id=0)
    with pytest.raises(RuntimeError, match="no encoded frame"):  # propagate() before encode()
        m.propagate()
    # state_dict() holds tensors under the reference's keys: torch.save(model.state_dict()) round-trips like the reference's checkpoints
    spec = arch.model_spec("td2", 19, "resnet18")
    sd = weights.synth_state_dict(spec, 5, 9, 3)
    m2 = td2_psp50.td2_psp50(nclass=19, path_num=2, model_path=None, backbone="resnet18").eval()
    m2.load_state_dict(sd)
    out = m2.state_dict()
    assert list(out) == list(sd) and all(torch.is_tensor(v) for v in out.values())
    assert out["pretrained1.bn1.num_batches_tracked"].dtype == torch.int64 and out["pretrained1.conv1.weight"].dtype == torch.float32
    import io
    buf = io.BytesIO()
    torch.save(out, buf)
    buf.seek(0)
    back = torch.load(buf)
    assert all(np.array_equal(back[k].numpy(), np.asarray(sd[k])) for k in sd)
    # load_state_dict(strict=False): nn.Module's bookkeeping -- unexpected keys dropped, missing ones reported (not the unused
    # pretrainedN.fc.* / num_batches_tracked the library ignores anyway); strict=True keeps everything for the library's strict check
    part = {k: v for k, v in sd.items() if not k.startswith("enc2.")}
    part["not.in.the.reference"] = np.zeros(2, np.float32)
    res = m2.load_state_dict(part, strict=False)
    assert res.unexpected_keys == ["not.in.the.reference"]
    assert res.missing_keys and all(k.startswith("enc2.") and not k.endswith("num_batches_tracked") for k in res.missing_keys)
    assert "not.in.the.reference" not in m2.state_dict()
    res = m2.load_state_dict(sd, strict=False)
    assert res.missing_keys == [] and res.unexpected_keys == []
    res = m2.load_state_dict(part)                                # strict: kept as given; the C library rejects it when the handle is built
    assert "not.in.the.reference" in m2.state_dict()


def test_streams_share_queue_argument_checks():
    """include/tdnet.h tdnet_streams_share_queue: the two paths that need no device -- a NULL result pointer is an error with a message,
    a stream compared with itself shares its queue by definition (the spin-pair test itself is exercised by the GPU batch tests)."""
    lib = _capi.lib()
    assert lib.tdnet_streams_share_queue(None, None, None) != 0
    assert b"shared is NULL" in lib.tdnet_last_error()
    shared = ctypes.c_int(0)
    assert lib.tdnet_streams_share_queue(ctypes.c_void_p(64), ctypes.c_void_p(64), ctypes.byref(shared)) == 0 and shared.value == 1


def test_the_package_caps_the_hardware_queues_before_the_runtime_starts():
    """tdnet_amd/__init__.py: GPU_MAX_HW_QUEUES defaults to 2 (a handle created behind an RCCL communicator otherwise runs at 0.67x) unless the
    environment already says otherwise; bench.py, conftest.py and __graft_entry__.py set it before they import torch."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    code = "import os, sys; sys.path.insert(0, %r); import tdnet_amd; print(os.environ['GPU_MAX_HW_QUEUES'], '|', tdnet_amd.hw_queue_note())" % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out.startswith("2 |") and "set by tdnet_amd at import" in out, out
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="3"), stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out.startswith("3 |") and "from the environment" in out, out
    for f in ("bench.py", "__graft_entry__.py", os.path.join("tests", "conftest.py")):
        src = open(os.path.join(ROOT, f)).read()
        assert src.index('setdefault("GPU_MAX_HW_QUEUES"') < (src.index("import torch") if "import torch" in src else len(src)), f
