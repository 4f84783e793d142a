from . import qarv  # registers qarv_base (reference: lvae/models/__init__.py:1-3)
