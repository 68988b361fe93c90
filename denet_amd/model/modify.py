"""Model surgery — the operations of denet/model/modify.py (:37-193) that assemble a DeNet detector from a
classifier: --class-num, --image-size, --convert-bn-relu (:76-113), --modify-bn (:115-131), --layer-remove
(:153-156), --layer-insert N:DESC (:161-175), --layer-append (:177-184). Like the reference every edit goes
through the JSON form and a reload, so shapes and wiring are rebuilt from scratch."""
import copy

from . import model_cnn


def _reload(model, json_obj=None):
    json_obj = model.export_json() if json_obj is None else json_obj
    new = model_cnn.load_from_json(json_obj, model.batch_size)
    new.class_labels = model.class_labels
    return new


def set_class_num(model, class_num):
    j = model.export_json()
    j["classNum"] = class_num
    return _reload(model, j)


def set_image_size(model, width, height):
    j = model.export_json()
    j["dataShape"] = (3, height, width)
    return _reload(model, j)


def _bn_to_bnrelu(bn_json):
    out = {"type": "batchnorm-relu", "layers": []}
    for k in ("momentum", "eps", "mean", "std", "gamma", "bias"):
        out[k] = bn_json[k]
    return out


def convert_bn_relu(model):
    """fuse [batchnorm, activation(relu)] pairs, at top level and inside `original` residual blocks"""
    j = model.export_json()
    src = j["layers"]
    dst = []
    i = 0
    while i < len(src):
        l = src[i]
        nxt = src[i + 1] if i + 1 < len(src) else None
        if l["type"] == "batchnorm" and nxt is not None and nxt["type"] == "activation" and nxt["activation"] == "relu" \
                and i + 1 < len(src) - 0:
            dst.append(_bn_to_bnrelu(l))
            i += 2
            continue
        if l["type"] == "resnet" and "bnrelu" not in l["version"] and l["activation"] == "relu":
            # the reference rewrites sub-layers [2],[3] (and [4],[5] with a bottleneck) of `original` blocks
            # (modify.py:91-106); fusing every [batchnorm, activation] pair of the block is the same edit for
            # those and also covers pre-activation blocks (which the reference's index arithmetic does not)
            l = copy.copy(l)
            subs, fused, k = list(l["layers"]), [], 0
            while k < len(subs):
                nx = subs[k + 1] if k + 1 < len(subs) else None
                if subs[k]["type"] == "batchnorm" and nx is not None and nx["type"] == "activation" \
                        and nx["activation"] == "relu":
                    fused.append(_bn_to_bnrelu(subs[k]))
                    k += 2
                else:
                    fused.append(subs[k])
                    k += 1
            l["layers"] = fused
            l["version"] = l["version"] + ",bnrelu"
        dst.append(l)
        i += 1
    j["layers"] = dst
    return _reload(model, j)


def modify_bn(model, enabled, momentum, eps):
    j = model.export_json()
    upd = {"enabled": bool(enabled), "momentum": float(momentum), "eps": float(eps)}
    for l in j["layers"]:
        if l["type"] == "batchnorm":
            l.update(upd)
        elif l["type"] == "resnet":
            l["bnParam"] = dict(l.get("bnParam", {}), **upd)
            for s in l["layers"]:
                if s["type"] in ("batchnorm", "batchnorm-relu"):
                    s["momentum"], s["eps"] = upd["momentum"], upd["eps"]
    return _reload(model, j)


def layer_remove(model, n):
    if n <= 0:
        return model
    j = model.export_json()
    j["layers"] = j["layers"][:-n]
    return _reload(model, j)


def layer_insert(model, inserts, activation="relu", border_mode="half", weight_init="he-backward"):
    """inserts: ["N:DESC", ...]; N indexes the CURRENT layer list including the initial layer (modify.py:163-172)"""
    for s in inserts:
        index, desc = s.split(":")
        index = int(index)
        if index > len(model.layers):
            raise Exception("Error: index %i too large (%i layers)" % (index, len(model.layers)))
        before = list(model.layers[:index])
        after = list(model.layers[index:])
        n0 = len(before)
        model.build_layer(desc, before, activation, border_mode, weight_init)
        new_json = [l.export_json() for l in before[n0:]]
        j = model.export_json()
        j["layers"] = j["layers"][:index - 1] + new_json + j["layers"][index - 1:]
        model = _reload(model, j)
    return model


def layer_append(model, descs, activation="relu", border_mode="half", weight_init="he-backward"):
    if isinstance(descs, str):
        descs = descs.split()
    for desc in descs:
        model.build_layer(desc, model.layers, activation, border_mode, weight_init)
    model._packed = False
    return _reload(model)


def merge_splits(model):
    """--merge (modify.py:53-60): split points become plain pass-throughs (they already execute as such here)"""
    j = model.export_json()
    for l in j["layers"]:
        if l["type"] == "split":
            l["enabled"] = False
        elif l["type"] == "skip-src":
            l["split"] = False
    return _reload(model, j)


def set_activation(model, activation):
    """modify.py:48-51: every activation / residual layer takes the activation named on the command line"""
    j = model.export_json()
    changed = False
    for l in j["layers"]:
        if l["type"] in ("activation", "resnet") and l.get("activation") != activation:
            l["activation"] = activation
            changed = True
    return _reload(model, j) if changed else model


def modify_layer(model, layer_name, assignments):
    """--modify-layer TYPE key=value ...: set attributes of the FIRST layer of that type (modify.py:130-146); the value
    takes the type of the attribute it replaces. The edit goes through the layer's JSON keys and a reload."""
    for index, layer in enumerate(model.layers):
        if layer.type_name != layer_name:
            continue
        before = layer.export_json()
        for a in assignments:
            name, text = a.split("=")
            old = getattr(layer, name)
            value = {"True": True, "False": False, "0": False, "1": True}[text] if type(old) is bool else type(old)(text)
            setattr(layer, name, value)
        after = layer.export_json()
        if after == before:
            raise Exception("--modify-layer: none of %s is part of the exported state of '%s'" % (assignments, layer_name))
        j = model.export_json()
        j["layers"][index - 1] = after
        return _reload(model, j)
    raise Exception("--modify-layer: no layer of type '%s'" % layer_name)


def build_parser():
    import argparse
    parser = argparse.ArgumentParser(description="Modify a model file (same flags as the reference's model-modify)")
    from ..common import logging
    logging.add_arguments(parser)
    parser.add_argument("--seed", type=int, default=23455, help="Random Seed for weights")
    parser.add_argument("--input", type=str, required=True)
    parser.add_argument("--output", type=str, required=True)
    parser.add_argument("--class-num", type=int, default=None)
    parser.add_argument("--image-size", nargs="+", type=int, default=None)
    parser.add_argument("--use-cudnn-pool", default=False, action="store_true",
                        help="accepted for the recipes' sake: every pooling layer of this build already pools the cuDNN way")
    parser.add_argument("--convert-bn-relu", default=False, action="store_true")
    parser.add_argument("--merge", default=False, action="store_true", help="merge split layers")
    parser.add_argument("--modify-bn", default=None, nargs="+", type=str, help="enabled momentum eps for batch norm")
    parser.add_argument("--modify-layer", default=None, nargs="+", type=str, help="TYPE key=value ...")
    parser.add_argument("--layer-insert", default=[], nargs="+", help="insert layer at position N:DESC")
    parser.add_argument("--layer-remove", default=0, type=int, help="remove N layer from end")
    parser.add_argument("--layer-append", default=[], nargs="+", type=str, help="append layers to end")
    parser.add_argument("--border-mode", default="half")
    parser.add_argument("--activation", default="relu")
    parser.add_argument("--weight-init", nargs="+", default=["he-backward"])
    return parser


def main(argv=None):
    """the edits of the reference's model-modify (modify.py:37-193)"""
    import random
    import numpy
    args = build_parser().parse_args(argv)
    random.seed(args.seed)
    numpy.random.seed(args.seed)
    model = model_cnn.load_from_file(args.input)
    # the reference edits the loaded object and reloads ONCE (modify.py:153-159); here every edit reloads, so the ones
    # that drop layers run before the ones that change the input geometry
    if args.modify_bn is not None:
        model = modify_bn(model, bool(args.modify_bn[0]), float(args.modify_bn[1]), float(args.modify_bn[2]))
    if args.convert_bn_relu:
        model = convert_bn_relu(model)
    model = layer_remove(model, args.layer_remove)
    if args.class_num is not None:
        model = set_class_num(model, args.class_num)
    if args.image_size is not None:
        model = set_image_size(model, args.image_size[0], args.image_size[1])
    model = set_activation(model, args.activation)
    if args.merge:
        model = merge_splits(model)
    if args.modify_layer is not None:
        model = modify_layer(model, args.modify_layer[0], args.modify_layer[1:])
    if len(args.layer_insert) > 0:
        model = layer_insert(model, args.layer_insert, args.activation, args.border_mode, args.weight_init)
    if len(args.layer_append) > 0:
        model = layer_append(model, args.layer_append, args.activation, args.border_mode, args.weight_init)
    model_cnn.save_to_file(model, args.output)
    for layer in model.layers:
        print(layer)
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
