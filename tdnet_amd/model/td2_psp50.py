"""`td2_psp50.td2_psp50(nclass=19, path_num=2, model_path=..., backbone='resnet18'|'resnet34')` -- Testing/test.py:32.

Mirror of Testing/model/pspnet/td2_psp50.py:29-155 (2 sub-networks, FIFO depth 1, d_v = 128).  The class keeps the
reference's name; the shipped ResNet-50 variant is SURVEY.md §8f "next" (N1) and raises NotImplementedError for now,
so the default backbone here is the BasicBlock one BASELINE.json's configs use (td2-psp18).
"""
from ._base import _TDNetBase


class td2_psp50(_TDNetBase):
    _model_id = 2
    _spec_name = "td2"

    def __init__(self, nclass=21, norm_layer=None, backbone="resnet18", dilated=True, aux=True, multi_grid=True,
                 path_num=None, model_path=None, synthetic_seed=None):
        super().__init__(nclass, norm_layer, backbone, dilated, aux, multi_grid, path_num, model_path, synthetic_seed)
