from ppsurf_amd.lightning_api import PPSurfModel  # noqa: F401
from ppsurf_amd.modules import PPSurfNetwork  # noqa: F401
