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
  * src/circuit_analyzer.py:937-1284  CircuitAnalyzer.crop_image_and_adjust_bboxes (+ :892-935 helpers), called
                           unbound on an object that carries only `debug` and `non_components` (:51)  -> crop.json

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
    _stub("ultralytics", "cv2", "dotenv", "google", "google.genai", "google.genai.types", "streamlit", "openai",
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


def import_analyzer():
    """src/circuit_analyzer.py (after import_reference()): the class only; its constructor loads model files and is never run."""
    return importlib.import_module("src.circuit_analyzer").CircuitAnalyzer


COMPONENT_CLASSES = ["resistor", "capacitor.unpolarized", "inductor", "diode", "voltage.dc", "gnd", "terminal", "transistor.bjt"]


def _box(cls, x0, y0, w, h, i, W, H, conf=0.5):
    x0, y0 = max(0, min(W - 1, int(x0))), max(0, min(H - 1, int(y0)))
    x1, y1 = max(x0, min(W, x0 + int(w))), max(y0, min(H, y0 + int(h)))
    return {"class": cls, "confidence": conf, "xmin": x0, "ymin": y0, "xmax": x1, "ymax": y1,
            "persistent_uid": f"{cls}_{x0}_{y0}_{x1}_{y1}_{i}"}


def schematic_boxes(rng, W, H, groups, n_text, n_far_text=0, n_junction=0, n_other=0, size=(20, 90)):
    """Detections of a drawn circuit: `groups` = [(centre x, centre y, spread, count)] of component boxes, text labels next to random
    components (and some far away), junction dots, and classes the clustering ignores (explanatory / circuit / vss / crossover)."""
    out = []
    for cx, cy, spread, count in groups:
        for _ in range(count):
            w, h = rng.integers(size[0], size[1], size=2)
            x, y = cx + rng.integers(-spread, spread + 1), cy + rng.integers(-spread, spread + 1)
            out.append(_box(COMPONENT_CLASSES[int(rng.integers(0, len(COMPONENT_CLASSES)))], x, y, w, h, len(out), W, H, float(rng.random())))
    comps = list(out)
    for _ in range(n_junction):
        b = comps[int(rng.integers(0, len(comps)))] if comps else None
        x, y = (b["xmin"] + rng.integers(-60, 60), b["ymin"] + rng.integers(-60, 60)) if b else rng.integers(0, min(W, H), size=2)
        out.append(_box("junction", x, y, rng.integers(4, 12), rng.integers(4, 12), len(out), W, H, float(rng.random())))
    for _ in range(n_text):
        b = comps[int(rng.integers(0, len(comps)))] if comps else None
        x, y = (b["xmax"] + rng.integers(-10, 60), b["ymin"] + rng.integers(-50, 50)) if b else rng.integers(0, min(W, H), size=2)
        out.append(_box("text", x, y, rng.integers(15, 80), rng.integers(8, 30), len(out), W, H, float(rng.random())))
    for _ in range(n_far_text):
        out.append(_box("text", rng.integers(0, W), rng.integers(0, H), rng.integers(15, 80), rng.integers(8, 30), len(out), W, H, float(rng.random())))
    for k in range(n_other):
        out.append(_box(["explanatory", "circuit", "vss", "crossover"][k % 4], rng.integers(0, W), rng.integers(0, H), rng.integers(20, 300),
                        rng.integers(20, 300), len(out), W, H, float(rng.random())))
    order = rng.permutation(len(out))          # detections arrive sorted by confidence, not by kind
    return [out[int(i)] for i in order]


def crop_cases(rng):
    C = []
    def add(name, W, H, boxes, padding=80):
        C.append({"name": name, "width": W, "height": H, "padding": padding, "boxes": boxes})
    add("empty", 640, 480, [])
    add("text_only", 640, 480, schematic_boxes(rng, 640, 480, [], 5))
    add("ignored_classes_only", 800, 600, schematic_boxes(rng, 800, 600, [], 2, n_other=6))
    add("one_component", 1000, 800, schematic_boxes(rng, 1000, 800, [(400, 300, 0, 1)], 0))
    add("one_component_with_text", 1000, 800, schematic_boxes(rng, 1000, 800, [(400, 300, 0, 1)], 2))
    add("junctions_only", 900, 700, [_box("junction", 300 + 40 * i, 200 + 25 * (i % 3), 8, 8, i, 900, 700) for i in range(7)])
    add("junctions_only_with_text", 900, 700, [_box("junction", 300 + 30 * i, 200 + 25 * (i % 3), 8, 8, i, 900, 700) for i in range(5)]
        + [_box("text", 320, 190, 40, 14, 9, 900, 700)])
    for k in range(6):                                    # one dense circuit in a large page: the ordinary case
        W, H = int(rng.integers(900, 2400)), int(rng.integers(700, 1800))
        add(f"single_cluster_{k}", W, H, schematic_boxes(rng, W, H, [(W // 2, H // 2, min(W, H) // 5, int(rng.integers(6, 40)))],
                                                          int(rng.integers(0, 15)), n_far_text=int(rng.integers(0, 4)), n_junction=int(rng.integers(0, 10)),
                                                          n_other=int(rng.integers(0, 3))))
    for k in range(6):                                    # two or three separated groups: the score decides
        W, H = int(rng.integers(1600, 3000)), int(rng.integers(1200, 2200))
        groups = [(W // 5, H // 4, 90, int(rng.integers(2, 12))), (4 * W // 5, 3 * H // 4, 90, int(rng.integers(2, 12)))]
        if k % 2:
            groups.append((W // 2, H // 8, 60, int(rng.integers(1, 6))))
        add(f"multi_cluster_{k}", W, H, schematic_boxes(rng, W, H, groups, int(rng.integers(0, 8)), n_far_text=int(rng.integers(0, 5)),
                                                         n_junction=int(rng.integers(0, 8)), n_other=k % 3))
    for k in range(3):                                    # no text at all: the largest cluster wins (ties: the first)
        W, H = 2000, 1500
        add(f"no_text_equal_clusters_{k}", W, H, schematic_boxes(rng, W, H, [(300, 300, 40, 4), (1600, 1100, 40, 4), (1000, 200, 30, 3 + k)], 0))
    add("fills_the_page", 640, 480, schematic_boxes(rng, 640, 480, [(60, 60, 30, 3), (560, 400, 30, 3), (300, 240, 200, 12)], 3, size=(30, 120)))
    add("basis_exactly_90_percent", 1000, 1000, [_box("resistor", 0, 0, 1000, 900, 0, 1000, 1000)])
    add("basis_just_above_90_percent", 1000, 1000, [_box("resistor", 0, 0, 1000, 901, 0, 1000, 1000)])
    add("text_chain_grows_window", 3000, 600, [_box("resistor", 200, 250, 80, 40, 0, 3000, 600)]
        + [_box("text", 420 + 190 * i, 260, 60, 20, 1 + i, 3000, 600) for i in range(9)])
    add("text_chain_reversed_order", 3000, 600, [_box("resistor", 200, 250, 80, 40, 0, 3000, 600)]
        + [_box("text", 420 + 190 * i, 260, 60, 20, 1 + i, 3000, 600) for i in reversed(range(9))])
    add("padding_0_degenerate_box", 800, 600, [_box("resistor", 300, 200, 0, 50, 0, 800, 600)], padding=0)
    add("padding_20_default", 1400, 1000, schematic_boxes(rng, 1400, 1000, [(700, 500, 150, 15)], 6, n_far_text=2), padding=20)
    add("boxes_outside_window_dropped", 2400, 1800, schematic_boxes(rng, 2400, 1800, [(600, 500, 120, 14)], 5, n_far_text=6, n_other=4)
        + [_box("capacitor.unpolarized", 2200, 1650, 60, 40, 99, 2400, 1800), _box("junction", 2300, 100, 8, 8, 98, 2400, 1800)])
    add("touching_edges_count_as_overlap", 1200, 900, [_box("resistor", 100, 100, 50, 50, 0, 1200, 900), _box("diode", 150, 150, 50, 50, 1, 1200, 900),
                                                       _box("inductor", 900, 700, 50, 50, 2, 1200, 900)])
    add("link_distance_boundary", 2000, 400, [_box("resistor", 100, 100, 30, 40, 0, 2000, 400), _box("resistor", 230, 100, 30, 40, 1, 2000, 400),
                                              _box("resistor", 361, 100, 30, 40, 2, 2000, 400)])
    W, H = 4000, 3000                                      # the detector's maximum: 300 boxes
    add("max_det_300", W, H, schematic_boxes(rng, W, H, [(1000, 800, 500, 120), (3000, 2200, 400, 80)], 60, n_far_text=20, n_junction=15, n_other=5))
    fl = schematic_boxes(rng, 1500, 1100, [(700, 500, 200, 10)], 4)
    for b in fl:                                           # un-rounded coordinates (a caller that skips bboxes()' round())
        for key in ("xmin", "ymin", "xmax", "ymax"):
            b[key] = float(b[key]) + float(rng.integers(0, 4)) * 0.25
    add("float_coordinates", 1500, 1100, fl)
    return C


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
    # ---------------- the crop between the stages (src/circuit_analyzer.py:937-1284; caller analysis_pipeline.py:177) ----------------
    Analyzer = import_analyzer()
    an = Analyzer.__new__(Analyzer)
    an.debug = False
    an.non_components = set(["text", "junction", "crossover", "vss", "explanatory", "circuit"])      # the constant of :51
    rng_c = np.random.default_rng(20250705)
    out_cases = []
    for case in crop_cases(rng_c):
        img = np.arange(case["height"] * case["width"] * 3, dtype=np.uint32).astype(np.uint8).reshape(case["height"], case["width"], 3)
        got_img, got_boxes, info = an.crop_image_and_adjust_bboxes(img, [dict(b) for b in case["boxes"]], padding=case["padding"])
        mc = info.get("main_cluster_info")
        exp = {"crop_applied": bool(info["crop_applied"]), "reason_for_no_crop": info["reason_for_no_crop"],
               "crop_decision_source": info["crop_decision_source"], "final_crop_window_abs": info["final_crop_window_abs"],
               "cropped_image_dims": info["cropped_image_dims"], "original_image_dims": info["original_image_dims"],
               "clustering_proximity_threshold": info["clustering_proximity_threshold"], "num_clusters_found": info["num_clusters_found"],
               "num_component_type_bboxes": info["num_component_type_bboxes"], "num_text_type_bboxes": info["num_text_type_bboxes"],
               "main_cluster_num_elements": mc.get("num_elements") if isinstance(mc, dict) else None,
               "main_cluster_example_uid": mc.get("example_uid") if isinstance(mc, dict) else None,
               "crop_basis_bbox_before_padding": info["crop_basis_bbox_before_padding"], "window_after_main_padding": info["window_after_main_padding"],
               "text_uids_that_expanded_crop": [t["uid"] for t in info["text_bboxes_that_expanded_crop"]],
               "image_shape": list(got_img.shape), "image_first_pixel": got_img.reshape(-1)[:3].tolist() if got_img.size else None,
               "boxes": [[b["persistent_uid"], b["xmin"], b["ymin"], b["xmax"], b["ymax"]] for b in got_boxes]}
        out_cases.append(dict(case, expected=json.loads(json.dumps(exp, default=lambda o: o.item() if hasattr(o, "item") else list(o)))))
    with open(os.path.join(OUT, "crop.json"), "w") as f:
        json.dump({"source": "src/circuit_analyzer.py:937-1284 (+ :892-935), called as analysis_pipeline.py:177 does", "cases": out_cases}, f)

    # sample circuit image shipped with the reference (static asset, data only; BASELINE config 1 input)
    import shutil
    shutil.copyfile(os.path.join(REF, "static", "images", "circuits_1.jpg"), os.path.join(OUT, "circuits_1.jpg"))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
