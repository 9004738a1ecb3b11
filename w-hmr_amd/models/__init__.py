"""Same exports as the reference's models/__init__.py:1-2 (hot-path members)."""
from .whmr import whmr_net, WHMR  # noqa: F401
from .smpl import SMPL  # noqa: F401
from .maf_extractor import MAF_Extractor  # noqa: F401
from .pose_vit import get_vitpose_encoder  # noqa: F401
from .hmr import hmr, HMR  # noqa: F401
