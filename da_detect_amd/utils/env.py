"""Site hook run before anything else is imported (reference: maskrcnn_benchmark/utils/env.py:7-37): when
TORCH_DETECTRON_ENV_MODULE names a python file, its `setup_environment()` is executed; otherwise nothing happens.
Like the reference, importing this module performs the setup once."""
import os

from .imports import import_file


def setup_custom_environment(custom_module_path):
    module = import_file("maskrcnn_benchmark.utils.env.custom_module", custom_module_path)
    hook = getattr(module, "setup_environment", None)
    assert callable(hook), ("Custom environment module defined in {} does not have the required callable attribute "
                            "'setup_environment'.").format(custom_module_path)
    hook()


def setup_environment():
    path = os.environ.get("TORCH_DETECTRON_ENV_MODULE")
    if path:
        setup_custom_environment(path)


setup_environment()
