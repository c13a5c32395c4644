"""da_detect_amd — MI355X-native training hot path of DA-Detect (domain-adaptive Faster R-CNN).

Layout: csrc/ (HIP kernels + C-ABI, include/dadet.h) -> _lib.py (ctypes) -> _C.py (tensor-level ops, the
stand-in for maskrcnn_benchmark._C) -> layers / structures / modeling / solver / engine mirroring the
reference's package so `build_detection_model(cfg)` and the DA YAMLs are drop-in.
`compat.install()` additionally registers the modules under the `maskrcnn_benchmark.*` names.
"""
import os as _os

# ROCm runtime knob, read when the HIP runtime initialises (i.e. it only takes effect when this package is imported
# before the process's first device call): kernel arguments go straight to device memory — lower launch latency
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
