"""gzip-JSON model files with base64 .npy blobs: the `.mdl.gz` format of the reference
(denet/common/json_util.py:8-48)."""
import base64
import gzip
import io
import json
import os

import numpy


def numpy_to_json(obj):
    if isinstance(obj, numpy.ndarray):
        bio = io.BytesIO()
        numpy.save(bio, obj)
        return {"__class__": "numpy.ndarray", "__value__": base64.b64encode(bio.getvalue()).decode()}
    if isinstance(obj, (numpy.floating,)):
        return float(obj)
    if isinstance(obj, (numpy.integer,)):
        return int(obj)
    raise TypeError(type(obj))


def numpy_from_json(obj):
    if obj.get("__class__") == "numpy.ndarray":
        return numpy.load(io.BytesIO(base64.b64decode(obj["__value__"])))
    return obj


def json_from_gz(fname):
    with gzip.open(fname, "rt") as f:
        return json.load(f, object_hook=numpy_from_json)


def json_to_gz(fname, json_obj, compresslevel=9):
    with gzip.open(fname, "wt", compresslevel=compresslevel) as f:
        json.dump(json_obj, f, indent=2, default=numpy_to_json)


def json_from_file(fname):
    if os.path.splitext(fname)[1] == ".gz":
        return json_from_gz(fname)
    with open(fname, "rt") as f:
        return json.load(f, object_hook=numpy_from_json)


def json_to_file(fname, json_obj):
    if os.path.splitext(fname)[1] == ".gz":
        return json_to_gz(fname, json_obj)
    with open(fname, "wt") as f:
        json.dump(json_obj, f, indent=2, default=numpy_to_json)
