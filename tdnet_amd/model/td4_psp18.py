"""`from model import td4_psp18; td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=...)` -- Testing/test.py:26.

Mirror of Testing/model/pspnet/td4_psp18.py:29-229 (4 sub-networks, FIFO of 3 cached frames, 12 attention modules).
"""
from ._base import _TDNetBase


class td4_psp18(_TDNetBase):
    _model_id = 4
    _spec_name = "td4"
