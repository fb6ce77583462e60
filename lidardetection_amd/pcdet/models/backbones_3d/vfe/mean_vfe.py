"""re-export (the encoders live in encoders.py)"""
from .encoders import MeanVFE  # noqa: F401
