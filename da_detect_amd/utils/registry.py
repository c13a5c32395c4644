"""name -> object registry with decorator registration (reference: maskrcnn_benchmark/utils/registry.py)."""


class Registry(dict):
    def register(self, name, module=None):
        if module is not None:
            assert name not in self
            self[name] = module
            return module

        def deco(fn):
            assert name not in self
            self[name] = fn
            return fn

        return deco
