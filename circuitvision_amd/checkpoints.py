"""Checkpoint readers for the three formats the reference loads (SURVEY.md 8(f)-3, section 5 "Checkpoint"):

  * ultralytics `.pt` (circuit_analyzer.py:45): a pickle of LIVE ultralytics classes
    ({'model': DetectionModel(...), ...}).  ultralytics is not a dependency here, so the pickle is read with a
    restricted unpickler that maps every `ultralytics.*` (and any other unknown) class to an inert stub and lets only
    torch tensor-rebuild helpers and plain containers through; tensors, `names` and the scale are then harvested by
    walking the stubs' `_modules / _parameters / _buffers` dictionaries.  No code from the checkpoint is executed.
  * SAM 2 base checkpoint {'model': state_dict} (sam2_infer.py:333 build_sam2) and the fine-tuned PEFT-keyed
    state_dict (circuit_analyzer.py:227-233): plain `torch.load(weights_only=True)`; see sam2.SamStateDictParams.
"""
import collections
import io
import pickle
import zipfile

import torch

_SAFE_PREFIXES = ("torch._utils", "torch.storage", "torch._tensor", "torch.serialization", "collections", "builtins", "numpy")
_SAFE_NAMES = {
    ("torch", "FloatStorage"), ("torch", "HalfStorage"), ("torch", "BFloat16Storage"), ("torch", "LongStorage"), ("torch", "IntStorage"),
    ("torch", "DoubleStorage"), ("torch", "BoolStorage"), ("torch", "ByteStorage"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.nn.parameter", "Parameter"), ("torch", "Tensor"), ("torch", "float32"), ("torch", "float16"),
}
_BLOCKED_BUILTINS = {"eval", "exec", "compile", "open", "__import__", "getattr", "setattr", "delattr", "input", "globals", "locals", "vars"}


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
        if (module, name) in _SAFE_NAMES or any(module == p or module.startswith(p + ".") for p in _SAFE_PREFIXES):
            if module == "builtins" and name in _BLOCKED_BUILTINS:
                raise pickle.UnpicklingError(f"blocked builtin {name}")
            mod = __import__(module, fromlist=[name])
            return getattr(mod, name)
        if module == "torch.nn.parameter" and name == "Parameter":
            return torch.nn.Parameter
        return _stub_class(module, name)           # ultralytics.*, torch.nn.modules.*, pathlib, ...: never executed


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
    """-> {'state_dict': {ultralytics key: tensor}, 'names': {int: str}, 'scale': 'n'|'s'|'m'|'l'|'x'}"""
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
    return {"state_dict": sd, "names": names, "scale": scale}


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
