"""re-export (the encoders live in encoders.py)"""
from .encoders import PFNLayer, PillarVFE  # noqa: F401
