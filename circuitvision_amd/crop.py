"""The crop between the two stages: which window of the image the segmenter sees, decided from the detector's boxes.

Host-side mirror of `CircuitAnalyzer.crop_image_and_adjust_bboxes` (/root/reference/src/circuit_analyzer.py:937-1284; helpers :892-935),
which `run_segmentation_and_cropping` calls with padding = 80 between the detector and the segmenter
(/root/reference/src/analysis_pipeline.py:177 -> :206).  Same name, arguments, return triple and `crop_debug_info` keys (the Streamlit page
reads them, /root/reference/app.py:559-595), so `CircuitAnalyzer` can forward to it; the reference runs this on the host too (<= 300 boxes).

What is MI355X-specific is what the window is used FOR: `crop_window` returns only the window and the shifted boxes -- no pixel is copied --
and `CircuitPipeline` hands the window to `cvmi_sam2_transform_rects` as a source rectangle of the u8 image that is already in HBM (the
detector put it there), so SAM 2's input is resized straight out of the sub-rectangle: no cropped host copy, no second H2D.

The arithmetic is vectorised over the box list (numpy): pairwise interval gaps as matrices, clusters by label propagation.
Results are pinned by tests/golden/crop.json (vectors produced by the reference's own method).
"""
from copy import deepcopy

import numpy as np

NON_COMPONENTS = frozenset(("text", "junction", "crossover", "vss", "explanatory", "circuit"))      # circuit_analyzer.py:51
_NOT_CLUSTERED = frozenset(("text", "explanatory", "circuit", "vss", "crossover"))                  # :982 (junctions are clustered)
TEXT_PADDING, TEXT_REACH = 20, 150                                                                  # :1192, :1201


def _coords(boxes):
    """[n, 4] xmin, ymin, xmax, ymax.  Python ints stay exact in float64 (pixel coordinates), floats pass through."""
    return np.array([[b["xmin"], b["ymin"], b["xmax"], b["ymax"]] for b in boxes], dtype=np.float64).reshape(-1, 4)


def _near(a, b, dist):
    """[len(a), len(b)] bool: boxes overlap (closed intervals) or both axis gaps are <= dist (:892-928)."""
    gx = np.maximum(np.maximum(b[None, :, 0] - a[:, None, 2], a[:, None, 0] - b[None, :, 2]), 0)
    gy = np.maximum(np.maximum(b[None, :, 1] - a[:, None, 3], a[:, None, 1] - b[None, :, 3]), 0)
    return (gx <= dist) & (gy <= dist)


def _components(adj):
    """Connected components of a symmetric bool matrix: label = lowest member index, by min-propagation to the fixed point."""
    n = adj.shape[0]
    label = np.arange(n)
    adj = adj | np.eye(n, dtype=bool)
    while True:
        new = np.where(adj, label[None, :], n).min(axis=1)
        new = new[new]                                       # pointer jumping
        if np.array_equal(new, label):
            return label
        label = new


def _num(v):
    """Box coordinates are python ints in the pipeline (bboxes() rounds them); give the caller's type back."""
    return int(v) if float(v).is_integer() else float(v)


def crop_window(bboxes, image_hw, padding=20):
    """-> (window (x0, y0, x1, y1) in pixels or None when no crop applies, crop_debug_info).  Touches no pixels."""
    H, W = int(image_hw[0]), int(image_hw[1])
    info = {"crop_applied": False, "reason_for_no_crop": None, "original_image_dims": (W, H), "num_total_yolo_bboxes": len(bboxes),
            "num_component_type_bboxes": sum(1 for b in bboxes if b.get("class") not in NON_COMPONENTS),
            "num_text_type_bboxes": 0, "clustering_proximity_threshold": None, "num_clusters_found": None, "main_cluster_info": None,
            "crop_decision_source": "unknown", "crop_basis_bbox_before_padding": None, "padding_value": padding,
            "window_after_main_padding": None, "text_bboxes_that_expanded_crop": [], "final_crop_window_abs": None,
            "cropped_image_dims": (W, H)}
    cls = [b.get("class") for b in bboxes]
    e_idx = [i for i, c in enumerate(cls) if c not in _NOT_CLUSTERED]
    t_idx = [i for i, c in enumerate(cls) if c == "text"]
    info["num_text_type_bboxes"] = len(t_idx)
    if not e_idx:
        info["reason_for_no_crop"], info["crop_decision_source"] = "no_elements_for_clustering", "no_crop_due_to_no_clustering_elements"
        return None, info
    xy = _coords(bboxes)
    E, T = xy[e_idx], xy[t_idx]
    junction = np.array([cls[i] == "junction" for i in e_idx])
    comp = ~junction
    if comp.any():
        S, mult, floor = E[comp], 2.0, 30
    else:
        S, mult, floor = E, 2.5, 20
    # (sum / count, as the reference: np.mean's pairwise summation may round differently on non-integer coordinates)
    mw, mh = float(sum((S[:, 2] - S[:, 0]).tolist())) / len(S), float(sum((S[:, 3] - S[:, 1]).tolist())) / len(S)
    diag = float(np.sqrt(mw ** 2 + mh ** 2))
    link = max(int(diag * mult), floor)
    info["clustering_proximity_threshold"] = link
    label = _components(_near(E, E, link))
    roots = np.unique(label)                                 # ascending = order of each cluster's lowest index
    info["num_clusters_found"] = len(roots)
    text_dist = max(int((diag if diag > 0 else 30) * 0.75), 25)
    has_text = _near(E, T, text_dist).any(axis=1) & comp if len(t_idx) else np.zeros(len(E), dtype=bool)
    size = np.array([(label == r).sum() for r in roots])
    with_text = np.array([has_text[label == r].sum() for r in roots])
    ncomp = np.array([comp[label == r].sum() for r in roots])
    # best = head of a stable descending sort by (with_text, size): the first index of the lexicographic maximum
    top = np.flatnonzero((with_text == with_text.max()))
    best = int(top[np.argmax(size[top])])
    if with_text[best] == 0 and ncomp[best] > 0:
        pick = int(np.argmax(size))                          # the first of the largest clusters
        info["crop_decision_source"] = "main_cluster_fallback_no_text_assoc_in_best_with_components"
    else:
        pick = best
        info["crop_decision_source"] = "main_yolo_cluster_scored_by_text_assoc"
    members = np.flatnonzero(label == roots[pick])
    info["main_cluster_info"] = {"num_elements": int(len(members)), "text_assoc_count": int(with_text[pick]),
                                 "score": (int(with_text[pick]), int(size[pick])), "id": pick,
                                 "example_uid": bboxes[e_idx[int(members[0])]].get("persistent_uid")}
    M = E[members]
    bx0, by0, bx1, by1 = M[:, 0].min(), M[:, 1].min(), M[:, 2].max(), M[:, 3].max()
    info["crop_basis_bbox_before_padding"] = (_num(bx0), _num(by0), _num(bx1), _num(by1))
    area = float(H * W)
    if area > 0 and (max(0.0, bx1 - bx0) * max(0.0, by1 - by0)) / area > 0.90:
        info["reason_for_no_crop"] = "crop_basis_bbox_too_large"
        return None, info
    x0, y0 = max(0.0, bx0 - padding), max(0.0, by0 - padding)
    x1, y1 = min(float(W), bx1 + padding), min(float(H), by1 + padding)
    info["window_after_main_padding"] = (int(round(x0)), int(round(y0)), int(round(x1)), int(round(y1)))
    for i, (tx0, ty0, tx1, ty1) in zip(t_idx, T.tolist()):   # sequential by definition: each text box sees the window the previous ones left
        if tx1 < x0 - TEXT_REACH or tx0 > x1 + TEXT_REACH or ty1 < y0 - TEXT_REACH or ty0 > y1 + TEXT_REACH:
            continue
        n = (min(x0, max(0.0, tx0 - TEXT_PADDING)), min(y0, max(0.0, ty0 - TEXT_PADDING)),
             max(x1, min(float(W), tx1 + TEXT_PADDING)), max(y1, min(float(H), ty1 + TEXT_PADDING)))
        if n != (x0, y0, x1, y1):
            b = bboxes[i]
            info["text_bboxes_that_expanded_crop"].append({"uid": b.get("persistent_uid"), "class": b.get("class"),
                                                           "coords_original": (b["xmin"], b["ymin"], b["xmax"], b["ymax"]),
                                                           "coords_text_box_abs": (tx0, ty0, tx1, ty1)})
        x0, y0, x1, y1 = n
    win = (max(0, int(round(x0))), max(0, int(round(y0))), min(W, int(round(x1))), min(H, int(round(y1))))
    info["final_crop_window_abs"] = win
    if win[0] >= win[2] or win[1] >= win[3]:
        info["reason_for_no_crop"] = "invalid_region_after_expansion"
        return None, info
    info["cropped_image_dims"] = (win[2] - win[0], win[3] - win[1])
    info["crop_applied"] = True
    return win, info


def _copy_box(b):
    """deepcopy(b) as the reference takes it (:1257), without its cost on the flat dicts bboxes() builds (strings and numbers only)."""
    for v in b.values():
        if isinstance(v, (list, dict, set, tuple, np.ndarray)):
            return deepcopy(b)
    return dict(b)


def adjust_bboxes(bboxes, window):
    """:1256-1277: every box in the window's coordinates, clipped to it; a box with no positive area left is dropped.  Copies (the
    persistent_uid travels with them)."""
    if window is None:
        return [_copy_box(b) for b in bboxes]
    x0, y0, x1, y1 = window
    w, h = x1 - x0, y1 - y0
    out = []
    for b in bboxes:
        nx0, ny0, nx1, ny1 = max(0, b["xmin"] - x0), max(0, b["ymin"] - y0), min(w, b["xmax"] - x0), min(h, b["ymax"] - y0)
        if nx1 > nx0 and ny1 > ny0:
            nb = _copy_box(b)
            nb["xmin"], nb["ymin"], nb["xmax"], nb["ymax"] = nx0, ny0, nx1, ny1
            out.append(nb)
    return out


def crop_image_and_adjust_bboxes(image_to_crop, all_yolo_bboxes_input, padding=20):
    """The reference's method as a function: -> (cropped image -- a VIEW of the input, as numpy slicing gives the reference --, adjusted
    bboxes, crop_debug_info); the original image and copies of the boxes when no crop applies."""
    window, info = crop_window(all_yolo_bboxes_input, image_to_crop.shape[:2], padding)
    boxes = adjust_bboxes(all_yolo_bboxes_input, window)
    if window is None:
        return image_to_crop, boxes, info
    return image_to_crop[window[1]:window[3], window[0]:window[2]], boxes, info
