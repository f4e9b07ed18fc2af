"""`td2_psp50.td2_psp50(nclass=19, path_num=2, model_path=..., backbone='resnet18'|'resnet34')` -- Testing/test.py:32.

Mirror of Testing/model/pspnet/td2_psp50.py:29-155 (2 sub-networks, FIFO depth 1, d_v = 512*expansion/4).  All three
backbones the reference accepts run on the HIP path: resnet50 (the shipped td2-psp50: Bottleneck + deep stem, d_v 512),
resnet18 (td2-psp18, BASELINE.json configs 1-2) and resnet34.  Default = the reference's default, resnet50.
"""
from ._base import _TDNetBase


class td2_psp50(_TDNetBase):
    _model_id = 2
    _spec_name = "td2"

    def __init__(self, nclass=21, norm_layer=None, backbone="resnet50", dilated=True, aux=True, multi_grid=True,
                 path_num=None, model_path=None, synthetic_seed=None, kernel_opts=None):
        super().__init__(nclass, norm_layer, backbone, dilated, aux, multi_grid, path_num, model_path, synthetic_seed, kernel_opts)
