from ppsurf_amd.lightning_api import PocoModel  # noqa: F401
from ppsurf_amd.modules import PocoNetwork, InterpAttentionKHeadsNet  # noqa: F401
