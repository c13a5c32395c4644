"""Environment report logged at start-up (reference: maskrcnn_benchmark/utils/collect_env.py:1-14): torch's own report
plus the Pillow version; here also the HIP device the kernels were built for."""
import PIL
from torch.utils.collect_env import get_pretty_env_info


def get_pil_version():
    return "\n        Pillow ({})".format(PIL.__version__)


def collect_env_info():
    env_str = get_pretty_env_info()
    env_str += get_pil_version()
    try:
        import torch
        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(0)
            env_str += "\n        HIP device 0: {} ({} CUs, {:.0f} GB); libdadet_hip.so built for gfx950".format(
                p.name, p.multi_processor_count, p.total_memory / 2 ** 30)
    except Exception:      # the report is informational
        pass
    return env_str
