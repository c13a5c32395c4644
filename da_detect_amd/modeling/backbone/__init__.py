from .backbone import build_backbone

__all__ = ["build_backbone"]
