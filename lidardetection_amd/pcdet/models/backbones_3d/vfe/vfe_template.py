"""re-export (the encoders live in encoders.py)"""
from .encoders import VFETemplate  # noqa: F401
