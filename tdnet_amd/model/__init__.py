"""`from model import td4_psp18, td2_psp50, pspnet` as in Testing/model/__init__.py:1-3."""
from . import td4_psp18  # noqa: F401
from . import td2_psp50  # noqa: F401
from . import pspnet  # noqa: F401
