"""Generates tests/golden/desc_fixtures.json by EXECUTING the reference's model-description parser in the build container:

    python tests/golden/make_desc_fixtures.py          # needs /root/reference; never runs on the GPU box

The operator surface of the hot path is the `TYPE.TAGS[args]` grammar: `ModelCNN.build_layer` (denet/model/model_cnn.py:122-146)
splits a token and offers it to the `parse_desc` of every class of the registry (denet/layer/layer_types.py:17-25) in order; the
first that accepts constructs its layer(s). All of that is plain Python, but the modules import Theano. As in
make_layer_method_fixtures.py the script takes the `ast` FunctionDef nodes of `build_layer` and of the 19 `parse_desc` methods,
compiles each node alone (no source text is stored) and runs them with every layer class NAME bound to a recorder: what is
written down is which constructor the reference calls for a token, with which arguments (named by the parameter names of the
reference's own `__init__`, read from the AST) - defaults, tag semantics, parameter types (convert_num), registry order, and the
error for an unknown token. tests/test_host.py replays the same tokens through the build's parser with ITS classes recorded
the same way."""
import ast
import glob
import json
import os
import sys
import types

REF = "/root/reference"
sys.path.insert(0, REF)
import denet.common as common  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# the shape the recorded stubs report as their output (some parse_desc read the channel count of the layer in front)
STUB_SHAPE = (2, 48, 20, 24)

TOKENS = ["C[64,3]", "C.B[64,7,2]", "C[32,1,1]", "C[16,5,2]", "C.B[100,1]", "DC[16,4,2]", "DC.B[8,3]", "DC[8,2,2]", "P[3,2,1]", "P[2]",
          "P.A[7]", "P.A[2,2]", "P[3,2]", "PI[2]", "PI[4]", "BN", "BN[0.99,0.0001]", "BN[0.9,1e-5,3,5,1000]", "BNA", "BNA[0.95]",
          "BNA[0.9,0.001]", "A", "D[0.5]", "D", "RSN[64,3]", "RSN.O[128,3,2]", "RSN[256,3,1,64]", "nRSN.O[3,64,3]", "nRSN.O[4,128,3,2]",
          "nRSN[2,256,3,2,64]", "CM[24]", "CM[28,0.5,0.1]", "B[4]", "B", "R", "R.TB", "R.C", "R.B", "SPLIT", "SKIP[1]", "SKIP[0]",
          "SKIPSRC[0]", "SKIPSRC.X[1]", "DNC[96,100]", "DNC.C[64]", "DNC[128,50,0.1]", "DNS[7,24,0.01,0.1]", "DNS.G[5,12,0.05,0.25,1,0.7]",
          "DNS[7,48,0.01,0.1]", "DND[0.5,1,1]", "DND.J[0.5,1,1]", "DND.JB[0.5,1,1]", "DND[0.5,2,0,0.5]", "DND.B[0.4,1,1]", "XX[1]",
          "[3]", "C.[8,3]"]


def class_nodes():
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "denet", "layer", "*.py"))):
        with open(path) as f:
            tree = ast.parse(f.read(), filename=path)
        for node in tree.body:
            if isinstance(node, ast.ClassDef):
                out[node.name] = (path, node)
    return out


def registry_order():
    """the class names of layer_types, in order, read from the two list expressions of layer_types.py (no execution)"""
    with open(os.path.join(REF, "denet", "layer", "layer_types.py")) as f:
        tree = ast.parse(f.read())
    names = []
    for node in tree.body:
        if isinstance(node, (ast.Assign, ast.AugAssign)) and isinstance(node.value, ast.List):
            target = node.targets[0] if isinstance(node, ast.Assign) else node.target
            if getattr(target, "id", None) == "layer_types":
                names += [e.id for e in node.value.elts]
    return names


def main():
    nodes = class_nodes()
    order = registry_order()
    log = []

    def recorder(cls_name):
        path, node = nodes[cls_name]
        init = [i for i in node.body if isinstance(i, ast.FunctionDef) and i.name == "__init__"][0]
        arg_names = [a.arg for a in init.args.args][1:]            # without self

        def construct(*args, **kwargs):
            named = {}
            for n, v in zip(arg_names, args):
                named[n] = v
            named.update(kwargs)
            named.pop("layers", None)
            log.append({"class": cls_name, "args": {k: (list(v) if isinstance(v, tuple) else v) for k, v in named.items()}})
            return types.SimpleNamespace(output_shape=STUB_SHAPE, type_name=cls_name)
        return construct

    ns_classes = {name: recorder(name) for name in nodes}
    registry = []
    for name in order:
        path, node = nodes[name]
        fn_node = [i for i in node.body if isinstance(i, ast.FunctionDef) and i.name == "parse_desc"][0]
        ns = dict(ns_classes)
        ns["common"] = common
        exec(compile(ast.Module(body=[fn_node], type_ignores=[]), path, "exec"), ns)
        registry.append(types.SimpleNamespace(parse_desc=ns["parse_desc"], name=name))

    mpath = os.path.join(REF, "denet", "model", "model_cnn.py")
    with open(mpath) as f:
        mtree = ast.parse(f.read(), filename=mpath)
    model_cls = [n for n in mtree.body if isinstance(n, ast.ClassDef) and n.name == "ModelCNN"][0]
    bl = [i for i in model_cls.body if isinstance(i, ast.FunctionDef) and i.name == "build_layer"][0]
    ns = {"common": common, "layer_types": registry}
    exec(compile(ast.Module(body=[bl], type_ignores=[]), mpath, "exec"), ns)
    build_layer = ns["build_layer"]

    cases = []
    for tok in TOKENS:
        layers = [types.SimpleNamespace(output_shape=STUB_SHAPE, type_name="initial")]
        del log[:]
        holder = types.SimpleNamespace(class_num=80)
        try:
            build_layer(holder, tok, layers, "relu", "half", "he-backward")
            cases.append({"token": tok, "calls": [dict(c) for c in log], "layers_appended": len(layers) - 1})
        except Exception as exc:
            cases.append({"token": tok, "error": type(exc).__name__, "calls": [dict(c) for c in log]})
    fix = {"reference": "denet/model/model_cnn.py:%d-%d ModelCNN.build_layer + parse_desc of %d registry classes (denet/layer/layer_types.py)"
                        % (bl.lineno, bl.end_lineno, len(order)),
           "registry_order": order, "stub_shape": list(STUB_SHAPE), "cases": cases}
    path = os.path.join(HERE, "desc_fixtures.json")
    with open(path, "w") as f:
        json.dump(fix, f, indent=0)
    print("wrote", path, len(cases), "tokens;", sum(1 for c in cases if "error" in c), "rejected")
    for c in cases[:6] + cases[-4:]:
        print(c)


if __name__ == "__main__":
    main()
