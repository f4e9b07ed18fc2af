"""`from model import td4_psp18, td2_psp50` as in Testing/model/__init__.py:1-3 (pspnet = psp101 baseline is out of scope)."""
from . import td4_psp18  # noqa: F401
from . import td2_psp50  # noqa: F401
