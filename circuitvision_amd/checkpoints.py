"""Checkpoint readers for the three formats the reference loads (SURVEY.md 8(f)-3, section 5 "Checkpoint"):

  * ultralytics `.pt` (circuit_analyzer.py:45): a pickle of LIVE ultralytics classes
    ({'model': DetectionModel(...), ...}).  ultralytics is not a dependency here, so the pickle is read with a
    restricted unpickler that maps every `ultralytics.*` (and any other unknown) class to an inert stub and lets only
    torch tensor-rebuild helpers and plain containers through; tensors, `names` and the scale are then harvested by
    walking the stubs' `_modules / _parameters / _buffers` dictionaries.  Only the exact (module, name) pairs of
    `_SAFE_NAMES` resolve to real objects; every other global the pickle names becomes an inert stub class.
  * SAM 2 base checkpoint {'model': state_dict} (sam2_infer.py:333 build_sam2) and the fine-tuned PEFT-keyed
    state_dict (circuit_analyzer.py:227-233): plain `torch.load(weights_only=True)`; see sam2.SamStateDictParams.
"""
import collections
import io
import pickle
import zipfile

import torch

# Exact (module, name) allow-list: everything a tensor-bearing checkpoint needs to be rebuilt and nothing that can run code.
# No module prefix is trusted as a whole -- `numpy.testing._private.utils.runstring` (exec) and
# `torch.storage._load_from_bytes` (an unrestricted nested torch.load) both live under innocent-looking trees.
_STORAGES = ("FloatStorage", "HalfStorage", "BFloat16Storage", "DoubleStorage", "LongStorage", "IntStorage", "ShortStorage",
             "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage")
_DTYPES = ("float32", "float16", "bfloat16", "float64", "int64", "int32", "int16", "int8", "uint8", "bool")
_SAFE_NAMES = (
    {("torch", n) for n in _STORAGES + _DTYPES + ("Size", "device", "dtype", "Tensor")}
    | {("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage")}
    | {("torch._utils", n) for n in ("_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_parameter", "_rebuild_parameter_with_state")}
    | {("torch.nn.parameter", "Parameter"), ("torch._tensor", "_rebuild_from_type_v2")}
    | {("collections", "OrderedDict"), ("collections", "defaultdict")}
    | {("builtins", n) for n in ("set", "frozenset", "dict", "list", "tuple", "slice", "complex", "int", "float", "bool", "str",
                                 "bytes", "bytearray", "range", "object")}
    | {(m, n) for m in ("numpy.core.multiarray", "numpy._core.multiarray") for n in ("scalar", "_reconstruct")}
    | {("numpy", "dtype"), ("numpy", "ndarray")}
)


class _Stub:
    """Inert stand-in for any class the checkpoint references (ultralytics modules, torch.nn layers, loss objects...)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[0], dict):
            self.__dict__.update(state[0])

    def __call__(self, *a, **k):
        return _Stub()


def _stub_class(module, name):
    return type(name, (_Stub,), {"__module__": module})


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE_NAMES:
            mod = __import__(module, fromlist=[name])
            return getattr(mod, name)
        return _stub_class(module, name)           # ultralytics.*, torch.nn.modules.*, pathlib, os, numpy.testing, ...: inert


class _RestrictedPickle:
    """pickle_module for torch.load: same interface as `pickle`, restricted Unpickler."""
    __name__ = "restricted_pickle"
    Unpickler = RestrictedUnpickler
    load = staticmethod(lambda f, **k: RestrictedUnpickler(f, **k).load())
    loads = staticmethod(lambda b, **k: RestrictedUnpickler(io.BytesIO(b), **k).load())
    dumps, dump, HIGHEST_PROTOCOL, PicklingError, UnpicklingError = pickle.dumps, pickle.dump, pickle.HIGHEST_PROTOCOL, pickle.PicklingError, pickle.UnpicklingError


def _walk(mod, prefix, out):
    d = getattr(mod, "__dict__", {})
    for kind in ("_parameters", "_buffers"):
        for k, v in (d.get(kind) or {}).items():
            if torch.is_tensor(v):
                out[f"{prefix}{k}"] = v.detach()
    for k, sub in (d.get("_modules") or {}).items():
        if sub is not None:
            _walk(sub, f"{prefix}{k}.", out)


def load_ultralytics_pt(path):
    """-> {'state_dict': {ultralytics key: tensor}, 'names': {int: str}, 'scale': 'n'|'s'|'m'|'l'|'x', 'imgsz': int | None}"""
    ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_RestrictedPickle)
    if isinstance(ck, dict) and "state_dict" in ck and "scale" in ck:           # already converted
        return ck
    model = ck.get("ema") or ck.get("model") if isinstance(ck, dict) else ck
    if model is None:
        raise ValueError("no 'model' / 'ema' entry in the checkpoint")
    sd = collections.OrderedDict()
    _walk(model, "", sd)
    if not sd:
        raise ValueError("no tensors found in the checkpoint's module tree")
    sd = collections.OrderedDict((k, v.float()) for k, v in sd.items())
    names = getattr(model, "names", None) or (ck.get("names") if isinstance(ck, dict) else None)
    if isinstance(names, (list, tuple)):
        names = dict(enumerate(names))
    if not names:
        raise ValueError("checkpoint holds no class names")
    names = {int(k): str(v) for k, v in names.items()}
    yaml_ = getattr(model, "yaml", None)
    scale = yaml_.get("scale") if isinstance(yaml_, dict) else None
    if scale not in ("n", "s", "m", "l", "x"):
        scale = infer_scale(sd)
    # ultralytics' Model._load keeps train_args['imgsz'] as the predict-time default (overrides survive _reset_ckpt_args)
    targs = ck.get("train_args") if isinstance(ck, dict) else None
    if not isinstance(targs, dict):
        targs = getattr(model, "args", None)
        targs = targs if isinstance(targs, dict) else getattr(targs, "__dict__", {})
    imgsz = targs.get("imgsz") if isinstance(targs, dict) else None
    if isinstance(imgsz, (list, tuple)):
        imgsz = max(imgsz)
    imgsz = int(imgsz) if isinstance(imgsz, (int, float)) and imgsz > 0 else None
    return {"state_dict": sd, "names": names, "scale": scale, "imgsz": imgsz}


def infer_scale(sd):
    """YOLO11 scale letter from the widths of layer 0 / layer 8 (yolo11.yaml scale table)."""
    c0 = sd["model.0.conv.weight"].shape[0]
    c8 = sd["model.8.cv2.conv.weight"].shape[0]
    table = {(16, 256): "n", (32, 512): "s", (64, 512): "ml", (96, 768): "x"}
    s = table.get((c0, c8))
    if s is None:
        raise ValueError(f"unrecognised YOLO11 widths ({c0}, {c8})")
    if s == "ml":                      # m and l share widths; l has depth multiple 1.0 -> two repeats in layer 2
        s = "l" if any(k.startswith("model.2.m.1.") for k in sd) else "m"
    return s
