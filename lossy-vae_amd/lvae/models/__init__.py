from . import qresvae  # registers qres34m   (reference: lvae/models/__init__.py:1-3)
from . import qarv     # registers qarv_base
