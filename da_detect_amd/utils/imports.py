"""Load a python module from an explicit file path (reference: maskrcnn_benchmark/utils/imports.py).  Used for
`cfg.PATHS_CATALOG` (the site's dataset / model catalog) and for TORCH_DETECTRON_ENV_MODULE."""
import sys
from importlib import util as _util


def import_file(module_name, file_path, make_importable=False):
    """execute `file_path` as module `module_name`; with make_importable it is also registered in sys.modules so that a
    later `import module_name` finds it"""
    spec = _util.spec_from_file_location(module_name, file_path)
    if spec is None or spec.loader is None:
        raise ImportError("cannot load %r from %r" % (module_name, file_path))
    mod = _util.module_from_spec(spec)
    if make_importable:
        sys.modules[module_name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        if make_importable:
            sys.modules.pop(module_name, None)
        raise
    return mod


def load_paths_catalog(file_path):
    """the PATHS_CATALOG module, executed ONCE per file: a `DatasetCatalog.register(...)` or an edit of
    `ModelCatalog`'s tables made at run time is then seen by every later make_data_loader / catalog:// lookup (a fresh
    execution per call handed each caller its own pristine classes).  An already imported
    `maskrcnn_benchmark.config.paths_catalog` backed by the same file is reused."""
    import os

    name = "maskrcnn_benchmark.config.paths_catalog"
    want = os.path.abspath(file_path)
    mod = sys.modules.get(name)
    if mod is not None and os.path.abspath(getattr(mod, "__file__", "") or "") == want:
        return mod
    return import_file(name, want, True)
