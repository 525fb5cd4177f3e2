"""The ultralytics `.pt` reader: a checkpoint pickled from LIVE classes of a (fake) `ultralytics` package must load with
that package absent, without executing anything from it, and yield the same tensors / names / scale."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

from circuitvision_amd.checkpoints import infer_scale, load_ultralytics_pt
from oracle.yolo11 import YOLO11, randomize_


def _fake_ultralytics_checkpoint(path, scale="n", nc=7):
    """Pickle an oracle YOLO11 re-homed into modules named like ultralytics' (so the pickle references them)."""
    pkg = types.ModuleType("ultralytics"); nnm = types.ModuleType("ultralytics.nn"); tasks = types.ModuleType("ultralytics.nn.tasks")
    mods = types.ModuleType("ultralytics.nn.modules")
    sys.modules.update({"ultralytics": pkg, "ultralytics.nn": nnm, "ultralytics.nn.tasks": tasks, "ultralytics.nn.modules": mods})
    try:
        model = randomize_(YOLO11(scale, nc), seed=5)
        moved = {}
        for m in model.modules():
            cls = type(m)
            if cls.__module__.startswith("oracle"):
                if cls not in moved:
                    moved[cls] = type(cls.__name__, (cls,), {"__module__": "ultralytics.nn.modules"})
                    setattr(mods, cls.__name__, moved[cls])
                m.__class__ = moved[cls]
        DetectionModel = type("DetectionModel", (type(model),), {"__module__": "ultralytics.nn.tasks"})
        tasks.DetectionModel = DetectionModel
        model.__class__ = DetectionModel
        model.names = {i: f"part.{i}" for i in range(nc)}
        model.yaml = {"scale": scale, "nc": nc}
        ref_sd = {k: v.clone() for k, v in model.state_dict().items()}
        torch.save({"model": model, "epoch": 3, "train_args": {"imgsz": 640}}, path)
        return ref_sd
    finally:
        for k in ("ultralytics", "ultralytics.nn", "ultralytics.nn.tasks", "ultralytics.nn.modules"):
            sys.modules.pop(k, None)


def test_reads_ultralytics_pickle_without_ultralytics(tmp_path):
    path = str(tmp_path / "best.pt")
    ref_sd = _fake_ultralytics_checkpoint(path)
    assert "ultralytics" not in sys.modules
    with pytest.raises(Exception):                       # the stock loaders cannot read it
        torch.load(path, map_location="cpu", weights_only=True)
    ck = load_ultralytics_pt(path)
    assert ck["scale"] == "n" and ck["names"][3] == "part.3" and len(ck["names"]) == 7
    assert set(ck["state_dict"]) == set(ref_sd)
    for k, v in ref_sd.items():
        assert torch.equal(ck["state_dict"][k], v.float()), k
    assert infer_scale(ck["state_dict"]) == "n"
    # and it feeds the product's weight walker unchanged
    from circuitvision_amd._lib import F32
    from circuitvision_amd.yolo11 import StateDictParams, Yolo11Weights
    Yolo11Weights("n", 7, StateDictParams(ck["state_dict"]), F32, device="cpu")


def test_restricted_unpickler_does_not_execute_payloads(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /tmp/cvmi_pwned",))
    path = str(tmp_path / "evil.pt")
    torch.save({"model": Evil()}, path)
    if os.path.exists("/tmp/cvmi_pwned"):
        os.remove("/tmp/cvmi_pwned")
    with pytest.raises(Exception):
        load_ultralytics_pt(path)
    assert not os.path.exists("/tmp/cvmi_pwned")


def _payload_file(tmp_path, name, reduce_fn, args):
    """A zip checkpoint whose 'model' entry pickles as `reduce_fn(*args)` (GLOBAL + REDUCE), written with the stock pickler."""
    class Payload:
        def __reduce__(self):
            return (reduce_fn, args)
    path = str(tmp_path / name)
    torch.save({"model": Payload()}, path)
    return path


def test_allow_list_blocks_submodule_gadgets(tmp_path):
    """prefix allow-lists let `numpy.testing._private.utils.runstring` (exec) and
    `torch.storage._load_from_bytes` (nested unrestricted torch.load) through.  Both must now resolve to inert stubs."""
    import io
    from numpy.testing._private.utils import runstring
    from circuitvision_amd.checkpoints import RestrictedUnpickler, _Stub
    marker = "/tmp/cvmi_pwned2"
    for fn, args in ((runstring, (f"open('{marker}','w').write('x')", {})),
                     (torch.storage._load_from_bytes, (b"not a checkpoint",)),
                     (eval, (f"open('{marker}','w').write('x')",)),
                     (getattr, ("abc", "upper"))):
        if os.path.exists(marker):
            os.remove(marker)
        path = _payload_file(tmp_path, "gadget.pt", fn, args)
        with pytest.raises(Exception):
            load_ultralytics_pt(path)               # the stub result holds no module tree -> ValueError, nothing ran
        assert not os.path.exists(marker), fn
    for mod, name in (("numpy.testing._private.utils", "runstring"), ("torch.storage", "_load_from_bytes"), ("torch.serialization", "load"),
                      ("builtins", "eval"), ("builtins", "getattr"), ("os", "system"), ("torch._utils", "_import_dotted_name")):
        cls = RestrictedUnpickler(io.BytesIO(b"")).find_class(mod, name)
        assert isinstance(cls, type) and issubclass(cls, _Stub), (mod, name)


def test_checkpoint_imgsz_is_returned(tmp_path):
    path = str(tmp_path / "best.pt")
    _fake_ultralytics_checkpoint(path)
    assert load_ultralytics_pt(path)["imgsz"] == 640
