"""source/occupancy_data_module.py by name: dataset layout helpers (:18-71) and the loaders of :174-253."""
from ppsurf_amd.data import in_file_is_dataset, get_set_files, read_shape_list, get_pc_file, load_shape_data_pc  # noqa: F401
from ppsurf_amd.lightning_api import get_results_dir  # noqa: F401
from ppsurf_amd.meshio import load_pts  # noqa: F401
