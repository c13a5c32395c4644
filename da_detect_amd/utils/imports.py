"""Load a python module from an explicit file path (reference: maskrcnn_benchmark/utils/imports.py:12-24).
Used for `cfg.PATHS_CATALOG` (the site's dataset catalog) and TORCH_DETECTRON_ENV_MODULE."""
import importlib.util
import sys


def import_file(module_name, file_path, make_importable=False):
    spec = importlib.util.spec_from_file_location(module_name, file_path)
    if spec is None or spec.loader is None:
        raise ImportError("cannot load %r from %r" % (module_name, file_path))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    if make_importable:
        sys.modules[module_name] = module
    return module
