"""`from model import pspnet; pspnet.pspnet(nclass=19, model_path=...)` -- Testing/test.py:34-38 (`--model psp101`).

Mirror of Testing/model/pspnet/pspnet.py:31-115: the stateless single-frame PSPNet the reference uses as its comparison
model (ResNet-101 dilated multi-grid backbone + PSPHead = full pyramid pooling, conv3x3 4096->512, classifier).  It reuses
the TDNet kernels (Bottleneck convs, pyramid pooling, head, upsample); `pos_id` is accepted and ignored like in the
reference (pspnet.py:73)."""
import torch

from ._base import _TDNetBase
from .. import arch


class pspnet(_TDNetBase):
    _model_id = 1
    _spec_name = "psp"

    def __init__(self, nclass=21, norm_layer=None, backbone="resnet101", dilated=True, aux=True, multi_grid=True,
                 model_path=None, synthetic_seed=None, kernel_opts=None):
        if backbone not in ("resnet50", "resnet101"):
            if backbone in ("resnet18", "resnet34"):
                raise NotImplementedError("PSPNet with a BasicBlock backbone is not a configuration the reference ships")
            raise RuntimeError("unknown backbone: {}".format(backbone))             # pspnet.py:65-66
        torch.nn.Module.__init__(self)
        self.psp_path = model_path
        self.path_num = 1
        self.nclass = nclass
        self.backbone = backbone
        self.synthetic_seed = synthetic_seed
        self.kernel_opts = dict(kernel_opts or {})
        self._pending_shape = None
        self.spec = arch.model_spec("psp", nclass, backbone)
        self._state = None
        self._engine = None
        self._engine_key = None
        self._init_batch_state()
        self.pretrained_mp_load()

    def forward(self, x, pos_id=None):
        return super().forward(x[-1:], 0)                                          # pspnet.py:74: x = x[-1:]

    def forward_labels(self, x, pos_id=None):
        return super().forward_labels(x[-1:], 0)
