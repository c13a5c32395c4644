"""Minimal yacs-compatible configuration node.

`yacs` is not installed in the target image, so the subset of `yacs.config.CfgNode` the reference relies on
(reference: maskrcnn_benchmark/config/defaults.py:3,21; tools/train_net_triplet.py:311-315) is provided
here: attribute access, `merge_from_file`, `merge_from_list`, `merge_from_other_cfg`, `clone`, `freeze` /
`defrost`, `dump`.  Semantics follow yacs: unknown keys are an error, values must keep the type of the
default (tuple <-> list and int -> float are coerced), nested dicts become nodes.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute protocol ---------------------------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError("Attempted to set %s to %s, but CfgNode is immutable" % (name, value))
        self[name] = value

    # yacs API ---------------------------------------------------------------------------------
    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_frozen(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.is_frozen()
        return out

    def merge_from_file(self, path):
        with open(path, "r") as f:
            data = yaml.safe_load(f) or {}
        self._merge_dict(data, [])

    def merge_from_other_cfg(self, other):
        self._merge_dict(other, [])

    def merge_from_list(self, opts):
        if len(opts) % 2 != 0:
            raise AssertionError("Override list has odd length: %s; it must be a list of pairs" % (opts,))
        for full_key, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent key: %s" % full_key)
                node = node[p]
            leaf = parts[-1]
            if leaf not in node:
                raise KeyError("Non-existent key: %s" % full_key)
            node[leaf] = _coerce(_decode(raw), node[leaf], full_key)

    def _merge_dict(self, data, path):
        for k, v in data.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: %s" % full)
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("%s must be a mapping" % full)
                self[k]._merge_dict(v, path + [k])
            else:
                self[k] = _coerce(_decode(v), self[k], full)

    def dump(self, **kwargs):
        def plain(n):
            if isinstance(n, CfgNode):
                return {k: plain(v) for k, v in n.items()}
            if isinstance(n, tuple):
                return list(n)
            return n

        return yaml.safe_dump(plain(self), **kwargs)

    def __repr__(self):
        return "CfgNode(%s)" % dict.__repr__(self)


def _decode(v):
    """yacs decodes strings such as "(600,)" or "True" with literal_eval and leaves plain strings alone"""
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def _coerce(new, old, key):
    if old is None or type(new) is type(old):
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    raise ValueError("Type mismatch (%s vs. %s) with values (%r vs. %r) for config key: %s" %
                     (type(old), type(new), old, new, key))
