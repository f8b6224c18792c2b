from ppsurf_amd.spatial import normalize_patches, get_pts_local_ps  # noqa: F401
from ppsurf_amd.data import PPSurfDataModule  # noqa: F401
