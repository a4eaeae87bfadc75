"""
JSON round trip of the observation vector (reference json_save_load.py:4-181).  Wire format:
ndarrays become {"__ndarray__": true, "data": nested lists, "shape": [...]}.  Unlike the
reference, I/O errors propagate instead of being printed and swallowed.
"""
import json

import numpy as np


def numpy_array_encoder(obj):
    if isinstance(obj, np.ndarray):
        return {"__ndarray__": True, "data": obj.tolist(), "shape": obj.shape}
    raise TypeError(f"json_save_load: cannot encode a {type(obj).__name__} (only ndarrays get a wire form; the rest must be plain JSON)")


def numpy_array_decoder(dct):
    if dct.get("__ndarray__"):
        return np.array(dct["data"]).reshape(dct["shape"])
    return dct


def save_object(obj, filename):
    with open(filename, "w") as f:
        json.dump(obj, f, default=numpy_array_encoder)


def load_object(filename):
    with open(filename, "r") as f:
        return json.load(f, object_hook=numpy_array_decoder)
