"""API-surface fixture — runs ONLY in the authoring container (needs /root/reference).

Records, from the SOURCE TEXT of the reference's training entry point tools/train_net_triplet.py (parsed with `ast`,
nothing is imported or executed — several of those modules need torchvision / timm / cv2, absent here):
  * every `from maskrcnn_benchmark.X import name` (and the timm import inside train()), with the kind of object the
    name is bound to in the reference tree and — for functions and classes — its parameter list;
  * every call the script makes on one of those names: positional-argument count and keyword names.
tests/test_api_surface.py then checks that, after da_detect_amd.compat.install(), every name resolves and every
recorded call binds to this package's signature.  The fixture holds names and parameter lists (an interface
description), no reference source text.

    python tests/golden/make_golden_api.py
"""
import ast
import json
import os

REF = "/root/reference"
SCRIPT = os.path.join(REF, "tools", "train_net_triplet.py")
HERE = os.path.dirname(os.path.abspath(__file__))


def module_file(mod):
    base = os.path.join(REF, *mod.split("."))
    if os.path.isdir(base):
        return os.path.join(base, "__init__.py")
    return base + ".py"


def find_def(mod, name, depth=0):
    """-> (kind, ast node, module where it is defined); follows `from .x import name` re-exports"""
    path = module_file(mod)
    if not os.path.exists(path) or depth > 4:
        return None
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):          # also finds definitions nested under `if` (utils/imports.py)
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name == name:
            return ("class" if isinstance(node, ast.ClassDef) else "function"), node, mod
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in node.targets):
            return "object", node, mod
        if isinstance(node, ast.ImportFrom) and any((a.asname or a.name) == name for a in node.names):
            name = [a.name for a in node.names if (a.asname or a.name) == name][0]      # `from .defaults import _C as cfg`
            pkg = mod if path.endswith("__init__.py") else mod.rsplit(".", 1)[0]
            parts = pkg.split(".")
            if node.level > 1:
                parts = parts[: len(parts) - (node.level - 1)]
            target = ".".join(parts + ([node.module] if node.module else [])) if node.level else node.module
            return find_def(target, name, depth + 1)
    return None


def params(fn):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    ndef = len(a.defaults)
    out = []
    for i, n in enumerate(pos):
        d = a.defaults[i - (len(pos) - ndef)] if i >= len(pos) - ndef else None
        out.append({"name": n, "default": ast.unparse(d) if d is not None else None})
    return {"params": out, "vararg": a.vararg.arg if a.vararg else None,
            "kwonly": [x.arg for x in a.kwonlyargs], "kwarg": a.kwarg.arg if a.kwarg else None}


def main():
    tree = ast.parse(open(SCRIPT).read())
    imports, names = [], {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in ("maskrcnn_benchmark", "timm"):
            for a in node.names:
                rec = {"module": node.module, "name": a.name, "line": node.lineno}
                if node.module.startswith("maskrcnn_benchmark"):
                    found = find_def(node.module, a.name)
                    assert found is not None, (node.module, a.name)
                    kind, d, where = found
                    rec.update(kind=kind, defined_in=where)
                    if kind == "function":
                        rec["signature"] = params(d)
                    elif kind == "class":
                        init = [n for n in d.body if isinstance(n, ast.FunctionDef) and n.name == "__init__"]
                        rec["signature"] = params(init[0]) if init else None
                        rec["methods"] = sorted(n.name for n in d.body if isinstance(n, ast.FunctionDef)
                                                and not n.name.startswith("_"))
                else:
                    rec["kind"] = "third_party"
                imports.append(rec)
                names[a.asname or a.name] = rec
    calls = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in names:
            if any(isinstance(a, ast.Starred) for a in node.args) or any(k.arg is None for k in node.keywords):
                continue
            calls.append({"name": node.func.id, "line": node.lineno, "positional": len(node.args),
                          "keywords": [k.arg for k in node.keywords]})
    # methods the script (and the trainer it hands them to) calls on the objects it builds
    attr_calls = sorted({(n.func.value.id, n.func.attr) for n in ast.walk(tree)
                         if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)
                         and isinstance(n.func.value, ast.Name) and n.func.value.id in ("cfg", "checkpointer", "logger")})
    out = {"script": "tools/train_net_triplet.py", "imports": imports, "calls": calls,
           "attribute_calls": [list(x) for x in attr_calls]}
    path = os.path.join(HERE, "reference_api_surface.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s: %d imports, %d calls" % (path, len(imports), len(calls)))
    for r in imports:
        print("  %-48s %-28s %s" % (r["module"], r["name"], r["kind"]))
    for c in calls:
        print("  call %-24s line %d: %d positional + %s" % (c["name"], c["line"], c["positional"], c["keywords"]))


if __name__ == "__main__":
    main()
