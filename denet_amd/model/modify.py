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
