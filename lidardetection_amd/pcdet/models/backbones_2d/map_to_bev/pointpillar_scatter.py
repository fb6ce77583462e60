"""re-export (the BEV map modules live in bev_maps.py)"""
from .bev_maps import PointPillarScatter  # noqa: F401
