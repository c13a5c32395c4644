from .cfgnode import CfgNode
from .defaults import _C as cfg, get_cfg_defaults

__all__ = ["cfg", "CfgNode", "get_cfg_defaults"]
