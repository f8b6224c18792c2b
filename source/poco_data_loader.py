from ppsurf_amd.spatial import sampling_quantized, get_fkaconv_ids, get_proj_ids, get_data_poco  # noqa: F401
from ppsurf_amd.data import PocoDataModule, PocoDataset, ReconstructionDataset as PocoReconstructionDataset  # noqa: F401
