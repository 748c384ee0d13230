"""Build-container script (the reference is never shipped): walks the syntax tree of the reference's CALLER --
opt/opt.py, opt/util/config_util.py -- and of its Python glue's extension call sites (svox2/svox2.py: `_C.<fn>(...)`),
reading them as text, and writes tests/golden/opt_py_surface.json: NAMES ONLY -- every `grid.<attr>` and
`svox2.<attr>` the training script touches, every call on the grid with its positional count and keyword names, the
constructor's keywords, the RenderOptions fields it sets, every `_C.<fn>` call site of svox2.py with its positional
count, and the argparse options (name, default) whose values flow into those calls.
tests/test_api_surface.py::test_opt_py_surface_* binds each of them against the mirror.

    python tests/golden/gen_opt_py_surface.py [/root/reference]
"""
import ast
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "opt_py_surface.json")


def parse(rel):
    return ast.parse(open(os.path.join(REF, rel)).read(), filename=rel)


def dotted(node):
    """a.b.c -> ['a', 'b', 'c'] (None when the chain does not start at a plain name)"""
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return parts[::-1]
    return None


def args_used(node):
    """names x of every `args.x` inside an expression"""
    out = set()
    for n in ast.walk(node):
        d = dotted(n) if isinstance(n, ast.Attribute) else None
        if d and d[0] == "args" and len(d) == 2:
            out.add(d[1])
    return sorted(out)


def guards_hold(call, parents, values):
    """Do the `if` tests around a call site hold under the option values?  True / False / None (a test that needs more
    than `args` to evaluate: unknown, read as reachable).  Evaluated here; no expression text is kept."""
    import argparse
    # the grid as opt.py:125-143 constructs it from the options: basis type by name, a background iff layers > 0
    consts = {}
    for node in parse("svox2/defs.py").body:
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Constant):
            consts[node.targets[0].id] = node.value.value
    grid = argparse.Namespace(
        basis_type=consts.get("BASIS_TYPE_" + str(values.get("basis_type", "sh")).upper()),
        surface_type=consts.get("SURFACE_TYPE_" + str(values.get("surface_type", "none")).upper()),
        use_background=(values.get("background_nlayers", 0) or 0) > 0)
    env = {"args": argparse.Namespace(**values), "grid": grid, "svox2": argparse.Namespace(**consts),
           "len": len, "min": min, "max": max, "abs": abs}
    # module-level flags the script derives from its options (USE_KERNEL = not args.nokernel, ...)
    root = call
    while id(root) in parents:
        root = parents[id(root)]
    for node in getattr(root, "body", []):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            try:
                env[node.targets[0].id] = eval(compile(ast.Expression(node.value), "<flag>", "eval"), {"__builtins__": {}}, env)
            except Exception:  # noqa: BLE001
                pass
    verdict = True
    node = call
    while id(node) in parents:
        par = parents[id(node)]
        if isinstance(par, (ast.If, ast.While, ast.IfExp)) and node is not par.test:
            in_body = any(node is b for b in (par.body if isinstance(par.body, list) else [par.body]))
            try:
                val = bool(eval(compile(ast.Expression(par.test), "<guard>", "eval"), {"__builtins__": {}}, env))
                if val != in_body:
                    return False
            except Exception:  # noqa: BLE001
                verdict = None
        node = par
    return verdict


def call_record(call, lineno):
    return dict(line=lineno, n_pos=len([a for a in call.args if not isinstance(a, ast.Starred)]),
                star=any(isinstance(a, ast.Starred) for a in call.args),
                keywords=[k.arg for k in call.keywords if k.arg is not None],
                kw_from_args={k.arg: args_used(k.value) for k in call.keywords if k.arg is not None and args_used(k.value)},
                pos_from_args=[args_used(a) for a in call.args])


def walk_caller(rel, defaults, live):
    tree = parse(rel)
    parents = {}
    for par in ast.walk(tree):
        for ch in ast.iter_child_nodes(par):
            parents[id(ch)] = par

    def rec_of(node):
        r = call_record(node, node.lineno)
        r["reached_with_parser_defaults"] = guards_hold(node, parents, defaults)
        r["reached_with_live_config"] = guards_hold(node, parents, live)
        return r

    grid_attrs, svox2_attrs, opt_fields, stored = {}, {}, set(), {}
    grid_calls, svox2_calls = {}, {}
    called = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Call):
            d = dotted(node.func)
            if d and d[0] == "grid" and len(d) == 2:
                grid_calls.setdefault(d[1], []).append(rec_of(node))
                called.add(id(node.func))
            elif d and d[0] == "svox2" and len(d) >= 2:
                svox2_calls.setdefault(".".join(d[1:]), []).append(rec_of(node))
                called.add(id(node.func))
            elif d and d[0] == "grid" and len(d) == 3 and d[1] == "_C":
                grid_calls.setdefault("_C." + d[2], []).append(rec_of(node))
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            d = dotted(node)
            if not d:
                continue
            if d[0] == "grid" and len(d) >= 2:
                grid_attrs.setdefault(d[1], set()).add(node.lineno)
                if len(d) == 2 and isinstance(node.ctx, ast.Store):
                    stored.setdefault(d[1], set()).add(node.lineno)
                if d[1] == "opt" and len(d) >= 3:
                    opt_fields.add(d[2])
            elif d[0] == "svox2" and len(d) >= 2:
                svox2_attrs.setdefault(d[1], set()).add(node.lineno)
    # svox2.__dict__['BASIS_TYPE_' + ...] idioms
    dyn = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Subscript):
            d = dotted(node.value)
            if d == ["svox2", "__dict__"]:
                dyn.append(dict(line=node.lineno, expr=ast.unparse(node.slice)))
    return dict(file=rel,
                grid_attrs={k: sorted(v) for k, v in sorted(grid_attrs.items())},
                grid_attrs_assigned={k: sorted(v) for k, v in sorted(stored.items())},
                svox2_attrs={k: sorted(v) for k, v in sorted(svox2_attrs.items())},
                render_option_fields=sorted(opt_fields),
                grid_calls={k: v for k, v in sorted(grid_calls.items())},
                svox2_calls={k: v for k, v in sorted(svox2_calls.items())},
                svox2_dict_lookups=dyn)


def walk_argparse(rels):
    """parser.add_argument('--name', ..., default=..., type=..., action=...)"""
    out = {}
    for rel in rels:
        for node in ast.walk(parse(rel)):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
                names = [a.value for a in node.args if isinstance(a, ast.Constant) and isinstance(a.value, str)]
                longs = [n for n in names if n.startswith("--")]
                if not longs:
                    continue
                kw = {k.arg: k.value for k in node.keywords}
                rec = dict(file=rel, line=node.lineno)
                if "default" in kw:
                    try:
                        rec["default"] = ast.literal_eval(kw["default"])
                    except Exception:  # noqa: BLE001
                        rec["default_expr"] = ast.unparse(kw["default"])
                if "action" in kw and isinstance(kw["action"], ast.Constant):
                    rec["action"] = kw["action"].value
                    if kw["action"].value == "store_true":
                        rec.setdefault("default", False)
                    if kw["action"].value == "store_false":
                        rec.setdefault("default", True)
                if "type" in kw:
                    rec["type"] = ast.unparse(kw["type"])
                if "dest" in kw and isinstance(kw["dest"], ast.Constant):
                    rec["dest"] = kw["dest"].value
                out[longs[0][2:]] = rec
    return out


def walk_extension_calls(rel):
    """`_C.<fn>(...)` call sites of the glue and the f-string dispatch through `_C.__dict__[...]`"""
    tree = parse(rel)
    calls, dispatch = {}, []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call):
            d = dotted(node.func)
            if d and len(d) == 2 and d[0] == "_C":
                calls.setdefault(d[1], []).append(dict(line=node.lineno, n_pos=len(node.args),
                                                       keywords=[k.arg for k in node.keywords if k.arg]))
        if isinstance(node, ast.Subscript):
            d = dotted(node.value)
            if d == ["_C", "__dict__"]:
                dispatch.append(dict(line=node.lineno, expr=ast.unparse(node.slice)))
    classes = sorted({d[1] for node in ast.walk(tree) if isinstance(node, ast.Attribute)
                      for d in [dotted(node)] if d and len(d) == 2 and d[0] == "_C" and d[1][:1].isupper()})
    return dict(file=rel, calls={k: v for k, v in sorted(calls.items())}, dict_dispatch=dispatch, spec_classes=classes)


def yaml_config(rel):
    """the shipped config: lines `name = <python literal>` (the file is called .yaml; config_util.py evaluates it)"""
    vals = {}
    for line in open(os.path.join(REF, rel)):
        line = line.split("#", 1)[0].strip()
        if "=" not in line:
            continue
        name, expr = [t.strip() for t in line.split("=", 1)]
        try:
            vals[name] = ast.literal_eval(expr)
        except Exception:  # noqa: BLE001
            vals[name] = dict(expr=expr)
    return dict(file=rel, values=vals)


def main():
    ap = walk_argparse(["opt/opt.py", "opt/util/config_util.py"])
    cfg = yaml_config("opt/configs/surface_cuda_syn.yaml")
    defaults = {(v.get("dest") or k): v["default"] for k, v in ap.items() if "default" in v}
    live = dict(defaults)
    live.update({k: v for k, v in cfg["values"].items() if not isinstance(v, dict)})
    doc = dict(
        note="names, positional counts, option defaults and reachability flags extracted from the reference's sources by "
             "tests/golden/gen_opt_py_surface.py; no source text",
        caller=walk_caller("opt/opt.py", defaults, live),
        argparse=ap,
        extension=walk_extension_calls("svox2/svox2.py"),
        live_config=cfg,
    )
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    c = doc["caller"]
    print(f"{OUT}: {len(c['grid_attrs'])} grid attributes, {sum(len(v) for v in c['grid_calls'].values())} calls on the grid, "
          f"{len(c['svox2_attrs'])} svox2 names, {len(doc['argparse'])} options, {len(doc['extension']['calls'])} extension functions")


if __name__ == "__main__":
    main()
