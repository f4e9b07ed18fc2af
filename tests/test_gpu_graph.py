"""GPU: a frame call only ENQUEUES (include/tdnet.h "Conventions") -- so one pos_id cycle of a clip can be captured into a hipGraph on the
caller's stream and replayed.  P consecutive tdnet_forward / tdnet_forward_labels calls in steady state (one period of the K/Q/V ring,
td4_psp18.py:123-134) are captured with torch.cuda.CUDAGraph (tdnet_amd/graph.py GraphedClip) and replayed for the following cycles; every
replayed frame must equal the eager loop's (Testing/test.py:45-59) bit for bit: the ring rotates on the device exactly as eager mode
rotates it, the internal streams (cache-only attention chain, second row-parity chain) join the capture through their fork / join events."""
import pytest
import torch

from tdnet_amd import weights
from tdnet_amd.graph import GraphedClip
from tdnet_amd.model import td2_psp50, td4_psp18

pytestmark = pytest.mark.gpu


def _model(name, bb, opts):
    if name == "td4":
        return td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, backbone=bb, synthetic_seed=0, kernel_opts=opts).eval().to("cuda")
    return td2_psp50.td2_psp50(nclass=19, path_num=2, model_path=None, backbone=bb, synthetic_seed=0, kernel_opts=opts).eval().to("cuda")


def _graph_equals_eager(name, bb, H, W, opts, labels, cycles=3):
    P = 4 if name == "td4" else 2
    warm = 2 * P                                                       # steady state, and a whole number of cycles
    T = warm + cycles * P
    clip = [torch.from_numpy(x).cuda() for x in weights.synth_video(H, W, T, seed=5)]
    with torch.no_grad():
        m = _model(name, bb, opts)
        call = m.forward_labels if labels else m.forward
        eager = [call(clip[t], pos_id=t % P).clone() for t in range(T)]
        n_eager = m.engine.last_launch_count()
        m.reset()
        for t in range(warm):
            assert torch.equal(call(clip[t], pos_id=t % P), eager[t])
        g = GraphedClip(m, H, W, "cuda", labels=labels)                # the capture itself advances the host-side ring by one whole period
        for c in range(cycles):
            t0 = warm + c * P
            outs = g.replay(clip[t0:t0 + P])
            torch.cuda.synchronize()
            for j in range(P):
                assert torch.equal(outs[j], eager[t0 + j]), (name, bb, H, W, opts, labels, "cycle", c, "frame", j)
        # ... and the handle is still usable eagerly afterwards, in step with the clip
        m.reset()
        for t in range(P + 1):
            assert torch.equal(call(clip[t], pos_id=t % P), eager[t])
    assert n_eager > 20
    m.engine.close()


def test_td4_cycle_captured_and_replayed_fp32():
    _graph_equals_eager("td4", "resnet18", 129, 257, None, labels=False)


def test_td4_cycle_with_row_parity_chains_and_labels():
    """overlap bit 4 forces the two-stream row-parity chains at this size: three internal streams inside the capture"""
    _graph_equals_eager("td4", "resnet18", 129, 257, {"overlap": 41 | 4}, labels=True)


def test_td2_cycle_fp16_mode():
    _graph_equals_eager("td2", "resnet34", 97, 129, {"precision": 1}, labels=False)


def test_td4_cycle_headline_size_fp32():
    _graph_equals_eager("td4", "resnet18", 1024, 2048, None, labels=True, cycles=2)
