from ppsurf_amd.spatial import normalize_patches, get_pts_local_ps  # noqa: F401
from ppsurf_amd.data import PPSurfDataModule, PPSurfDataset, ReconstructionDataset as PPSurfReconstructionDataset  # noqa: F401
