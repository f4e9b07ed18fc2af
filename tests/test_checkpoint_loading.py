"""N2: a checkpoint written exactly like the reference's (torch.save of the nn.Module state_dict, incl. the unused
pretrainedN.fc.* and the int64 *.num_batches_tracked entries) goes through model_path / pretrained_mp_load
(td4_psp18.py:232-240).  CPU part: file -> host state; the GPU part of the same path is tests/test_gpu_harness.py."""
import numpy as np
import pytest
import torch

from tdnet_amd import arch, weights
from tdnet_amd.model import td2_psp50, td4_psp18


def _ckpt(tmp_path, spec, h, w):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in weights.synth_state_dict(spec, h, w, 0).items()}
    assert sd["pretrained1.bn1.num_batches_tracked"].dtype == torch.int64
    p = tmp_path / "ckpt.pkl"
    torch.save(sd, str(p))
    return str(p), sd


def test_model_path_loading_td4(tmp_path, capsys):
    spec = arch.model_spec("td4", 19, "resnet18")
    path, sd = _ckpt(tmp_path, spec, 97, 193)
    m = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=path)
    assert "Loading pretrained model from" in capsys.readouterr().out          # same message as td4_psp18.py:235
    got = m.state_dict()
    assert set(got) == set(sd) and len(got) == 728
    assert np.array_equal(got["head3.conv5.4.bias"], sd["head3.conv5.4.bias"].numpy())


def test_model_path_loading_td2_psp50(tmp_path):
    spec = arch.model_spec("td2", 19, "resnet50")
    path, sd = _ckpt(tmp_path, spec, 97, 193)
    m = td2_psp50.td2_psp50(nclass=19, path_num=2, model_path=path)          # default backbone = resnet50, as the reference
    assert set(m.state_dict()) == set(sd) and len(sd) == 776


def test_missing_checkpoint_is_an_error_unless_synthetic():
    with pytest.raises(FileNotFoundError):
        td4_psp18.td4_psp18(nclass=19, path_num=4, model_path="/nonexistent.pkl")
    td4_psp18.td4_psp18(nclass=19, path_num=4, model_path="/nonexistent.pkl", synthetic_seed=3)   # explicit opt-in
