"""Generate golden vectors from the reference's OWN code (run in the build container only).

The reference (/root/reference) cannot be imported as-is here: its module-level imports pull
in packages that are not installed (cv2, streamlit, torchvision, sam2, peft ...).  None of
those packages are used by the three functions captured below, so this script installs inert
stub modules for them in sys.modules, imports the reference modules, and records
input/expected-output pairs as small .npz / .json fixtures:

  * src/utils.py:297-361   calculate_iou, non_max_suppression_by_confidence,
                           non_max_suppression_by_area            -> nms_stage2.json
  * src/sam2_infer.py:130-189  MultiKernelRefinement.forward       -> refinement.npz
  * src/sam2_infer.py:88-128   SAM2Transforms.postprocess_masks     -> postprocess.npz
  * src/sam2_infer.py:250      learned dense prompt product         -> dense_prompt.npz

The fixtures are DATA (inputs + expected outputs); no reference source text is stored.
Usage:  python tests/golden/make_golden.py   (needs /root/reference; never runs on the GPU box)
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Anything(types.ModuleType):
    """Module stub: any attribute is a dummy class / callable."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        val = type(name, (), {"__init__": lambda self, *a, **k: None,
                              "__call__": lambda self, *a, **k: None})
        setattr(self, name, val)
        return val


def _stub(*names):
    for n in names:
        parts = n.split(".")
        for i in range(1, len(parts) + 1):
            key = ".".join(parts[:i])
            if key not in sys.modules:
                sys.modules[key] = _Anything(key)
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], sys.modules[key])


def import_reference():
    _stub("cv2", "dotenv", "google", "google.genai", "google.genai.types", "streamlit", "openai",
          "matplotlib", "matplotlib.pyplot", "torchvision", "torchvision.transforms",
          "sam2", "sam2.build_sam", "sam2.sam2_image_predictor", "sam2.modeling",
          "sam2.modeling.sam2_base", "sam2.utils", "sam2.utils.misc", "peft")
    sys.modules["dotenv"].load_dotenv = lambda *a, **k: None
    sys.path.insert(0, REF)
    utils = importlib.import_module("src.utils")
    # sam2_infer builds torchvision transforms inside an nn.Sequential at construction time; the
    # stubs above are not nn.Modules, so SAM2Transforms is instantiated via __new__ below.
    sam2_infer = importlib.import_module("src.sam2_infer")
    return utils, sam2_infer


def rand_boxes(rng, n, wh=640, tie_conf=False, zero_area=False):
    out = []
    for i in range(n):
        x0, y0 = int(rng.integers(0, wh - 10)), int(rng.integers(0, wh - 10))
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        if zero_area and i % 7 == 0:
            w = 0
        conf = float(rng.random())
        if tie_conf and i % 3 == 0:
            conf = 0.5
        out.append({"class": f"c{int(rng.integers(0, 5))}", "confidence": conf,
                    "xmin": x0, "ymin": y0, "xmax": min(x0 + w, wh), "ymax": min(y0 + h, wh),
                    "persistent_uid": f"u{i}"})
    return out


def clustered_boxes(rng, n):
    """Boxes jittered around a few centres so that many pairs have high IoU."""
    out = []
    centres = [(100, 100, 60, 40), (300, 200, 80, 80), (500, 400, 30, 90), (320, 320, 200, 200)]
    for i in range(n):
        cx, cy, w, h = centres[i % len(centres)]
        j = rng.integers(-6, 7, size=4)
        x0, y0 = cx - w // 2 + int(j[0]), cy - h // 2 + int(j[1])
        out.append({"class": "r", "confidence": float(rng.random()),
                    "xmin": x0, "ymin": y0, "xmax": x0 + w + int(j[2]), "ymax": y0 + h + int(j[3]),
                    "persistent_uid": f"u{i}"})
    return out


def main():
    utils, sam2_infer = import_reference()
    rng = np.random.default_rng(20250704)

    # ---------------- stage-2 NMS (src/utils.py:297-361) ----------------
    cases = []
    specs = [("empty", []), ("one", rand_boxes(rng, 1)), ("two", rand_boxes(rng, 2)),
             ("rand50", rand_boxes(rng, 50)), ("rand300", rand_boxes(rng, 300)),
             ("ties", rand_boxes(rng, 60, tie_conf=True)),
             ("zero_area", rand_boxes(rng, 40, zero_area=True)),
             ("clustered", clustered_boxes(rng, 120)),
             # IoU exactly 0.6: inter 60, union 100  -> must be SUPPRESSED-not (kept iff iou < thr)
             ("iou_exact_0p6", [
                 {"class": "a", "confidence": 0.9, "xmin": 0, "ymin": 0, "xmax": 10, "ymax": 8, "persistent_uid": "u0"},
                 {"class": "a", "confidence": 0.8, "xmin": 0, "ymin": 2, "xmax": 10, "ymax": 10, "persistent_uid": "u1"},
                 {"class": "a", "confidence": 0.7, "xmin": 100, "ymin": 100, "xmax": 100, "ymax": 100, "persistent_uid": "u2"},
                 {"class": "a", "confidence": 0.6, "xmin": 100, "ymin": 100, "xmax": 100, "ymax": 100, "persistent_uid": "u3"}])]
    for name, boxes in specs:
        for thr in (0.6, 0.5):
            kept_c = utils.non_max_suppression_by_confidence([dict(b) for b in boxes], iou_threshold=thr)
            kept_a = utils.non_max_suppression_by_area([dict(b) for b in boxes], iou_threshold=thr)
            cases.append({"name": name, "iou_threshold": thr, "boxes": boxes,
                          "kept_by_confidence": [b["persistent_uid"] for b in kept_c],
                          "kept_by_area": [b["persistent_uid"] for b in kept_a]})
    iou_pairs = []
    bx = rand_boxes(rng, 40, zero_area=True)
    for i in range(0, 40, 2):
        iou_pairs.append({"a": bx[i], "b": bx[i + 1], "iou": utils.calculate_iou(bx[i], bx[i + 1])})
    with open(os.path.join(OUT, "nms_stage2.json"), "w") as f:
        json.dump({"source": "src/utils.py:297-361", "cases": cases, "iou_pairs": iou_pairs}, f)

    # ---------------- refinement head (src/sam2_infer.py:130-189) ----------------
    torch.manual_seed(1234)
    ref = sam2_infer.MultiKernelRefinement(in_channels=1, out_channels=1, kernel_sizes=[3, 5, 7, 11],
                                           intermediate_channels=4).eval()
    assert sum(p.numel() for p in ref.parameters()) == 849
    with torch.no_grad():
        for p in ref.parameters():          # non-trivial biases too
            p.copy_(torch.randn_like(p) * 0.3)
        x1 = torch.randn(1, 1, 64, 64) * 4
        x2 = torch.randn(2, 1, 96, 80) * 4
        y1, y2 = ref(x1), ref(x2)
    sd = {k.replace(".", "__"): v.numpy() for k, v in ref.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "refinement.npz"), x1=x1.numpy(), y1=y1.numpy(),
                        x2=x2.numpy(), y2=y2.numpy(), **sd)

    # ---------------- postprocess_masks (src/sam2_infer.py:88-128, areas = 0) ----------------
    tr = sam2_infer.SAM2Transforms.__new__(sam2_infer.SAM2Transforms)
    torch.nn.Module.__init__(tr)
    tr.resolution, tr.mask_threshold, tr.max_hole_area, tr.max_sprinkle_area = 1024, 0, 0, 0
    m = torch.randn(2, 1, 128, 128) * 5
    outs = {}
    for hw in ((97, 211), (128, 128), (300, 60)):
        outs[f"out_{hw[0]}x{hw[1]}"] = tr.postprocess_masks(m, hw).numpy()
    np.savez_compressed(os.path.join(OUT, "postprocess.npz"), masks=m.numpy(), **outs)

    # ---------------- learned dense prompt (src/sam2_infer.py:206-209, 250) ----------------
    torch.manual_seed(7)
    e1, e2 = torch.randn(1, 256, 4), torch.randn(1, 4, 64 * 64)
    dense = (e1 @ e2).view(1, 256, 64, 64)
    # keep the fixture small: every 4th channel, every 4th row/col of the 256x64x64 product
    np.savez_compressed(os.path.join(OUT, "dense_prompt.npz"), e1=e1.numpy(), e2=e2.numpy(),
                        dense_sub=dense[:, ::4, ::4, ::4].numpy())
    # sample circuit image shipped with the reference (static asset, data only; BASELINE config 1 input)
    import shutil
    shutil.copyfile(os.path.join(REF, "static", "images", "circuits_1.jpg"), os.path.join(OUT, "circuits_1.jpg"))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
