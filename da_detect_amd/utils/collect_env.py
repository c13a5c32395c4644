"""Environment report logged once at start-up by the entry points (reference: maskrcnn_benchmark/utils/collect_env.py):
torch's own environment summary, the Pillow version (the host-side image decoding / resizing library whose arithmetic
the device pipeline reproduces), and the HIP device the kernels run on."""


def _pillow_line():
    import PIL

    return "        Pillow ({})".format(PIL.__version__)


def _device_line():
    import torch

    if not torch.cuda.is_available():
        return None
    props = torch.cuda.get_device_properties(0)
    return "        HIP device 0: {} ({} CUs, {:.0f} GB); libdadet_hip.so is built for gfx950".format(
        props.name, props.multi_processor_count, props.total_memory / 2 ** 30)


def get_pil_version():
    return "\n" + _pillow_line()


def collect_env_info():
    from torch.utils.collect_env import get_pretty_env_info

    parts = [get_pretty_env_info(), _pillow_line()]
    try:
        dev = _device_line()
        if dev:
            parts.append(dev)
    except Exception:      # informational only
        pass
    return "\n".join(parts)
