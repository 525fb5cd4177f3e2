"""Oracle: the crop between the two stages, CPU, pure Python.  TEST INFRASTRUCTURE ONLY.

Restates `CircuitAnalyzer.crop_image_and_adjust_bboxes` (/root/reference/src/circuit_analyzer.py:937-1284; helpers
`_are_bboxes_proximal_for_clustering` :892-928, `_component_has_nearby_text` :930-935) as called by
`run_segmentation_and_cropping` (/root/reference/src/analysis_pipeline.py:177, padding = 80): the detector's boxes decide
which window of the image the segmenter sees.  Pure integer / float arithmetic on <= 300 boxes.

PINNED by tests/golden/crop.json -- vectors produced by the reference's own method (tests/golden/make_golden.py imports
src/circuit_analyzer.py with inert stubs for the absent wheels and calls it unbound on seeded box sets).

The algorithm, in the order the reference's outputs depend on:
  elements  = boxes whose class is not text / explanatory / circuit / vss / crossover        (junctions DO count)   :980-983
  sizing    = the non-junction elements, or all elements when only junctions exist                                  :1003-1021
              diag = hypot(mean width, mean height);  link distance = max(int(2.0 diag), 30)  (junction-only: max(int(2.5 diag), 20))
  clusters  = connected components of "boxes overlap, or their edge gaps are <= link distance on both axes",
              numbered by their lowest element index                                                                :1027-1052
  score     = (# non-junction members with a text box within max(int(0.75 diag'), 25), # members), diag' = diag or 30  :1066-1088
  main      = best score (first among equals); if it has non-junction members but none with text: the LARGEST cluster
              (first among equals)                                                                                   :1091-1139
  basis     = bounding box of main; no crop when it covers > 90 % of the image                                      :1155-1181
  window    = basis +- padding clipped to the image, then grown by every text box (+- 20) that is not farther than
              150 px from the CURRENT window, in list order                                                          :1184-1231
  result    = window rounded to ints and clipped; every box shifted by the window origin, clipped to it, dropped
              when no positive area is left                                                                          :1235-1282
"""
from copy import deepcopy
from math import sqrt

NOT_CLUSTERED = ("text", "explanatory", "circuit", "vss", "crossover")            # :982
NON_COMPONENTS = ("text", "junction", "crossover", "vss", "explanatory", "circuit")   # circuit_analyzer.py:51


def _gap(lo1, hi1, lo2, hi2):
    """Distance between two closed intervals (0 when they touch or overlap)."""
    if hi1 < lo2:
        return lo2 - hi1
    if lo1 > hi2:
        return lo1 - hi2
    return 0


def near(a, b, dist):
    """:892-928.  Overlap (closed intervals) counts as near; otherwise both axis gaps must be <= dist."""
    return _gap(a["xmin"], a["xmax"], b["xmin"], b["xmax"]) <= dist and _gap(a["ymin"], a["ymax"], b["ymin"], b["ymax"]) <= dist


def _find(parent, i):
    while parent[i] != i:
        parent[i] = parent[parent[i]]
        i = parent[i]
    return i


def crop_plan(boxes, height, width, padding=20):
    """-> dict(applied, reason, source, window (x0, y0, x1, y1) or None, link_distance, clusters, main_size, basis, padded,
    text_uids).  No image is touched."""
    plan = {"applied": False, "reason": None, "source": "unknown", "window": None, "link_distance": None, "clusters": None,
            "main_size": None, "basis": None, "padded": None, "text_uids": []}
    elems = [b for b in boxes if b.get("class") not in NOT_CLUSTERED]
    texts = [b for b in boxes if b.get("class") == "text"]
    if not elems:
        plan["reason"], plan["source"] = "no_elements_for_clustering", "no_crop_due_to_no_clustering_elements"
        return plan
    sizing = [b for b in elems if b.get("class") != "junction"]
    if sizing:
        mult, floor = 2.0, 30
    else:
        sizing, mult, floor = elems, 2.5, 20
    mean_w = sum(b["xmax"] - b["xmin"] for b in sizing) / len(sizing)
    mean_h = sum(b["ymax"] - b["ymin"] for b in sizing) / len(sizing)
    diag = sqrt(mean_w ** 2 + mean_h ** 2)
    link = max(int(diag * mult), floor)
    plan["link_distance"] = link
    n = len(elems)
    parent = list(range(n))
    for i in range(n):
        for j in range(i + 1, n):
            if near(elems[i], elems[j], link):
                ri, rj = _find(parent, i), _find(parent, j)
                if ri != rj:
                    parent[max(ri, rj)] = min(ri, rj)
    groups = {}
    for i in range(n):                       # dict insertion order = order of each cluster's lowest index
        groups.setdefault(_find(parent, i), []).append(i)
    clusters = list(groups.values())
    plan["clusters"] = len(clusters)
    text_dist = max(int((diag if diag > 0 else 30) * 0.75), 25)
    scored = []
    for members in clusters:
        comps = [elems[i] for i in members if elems[i].get("class") != "junction"]
        with_text = sum(1 for c in comps if any(near(c, t, text_dist) for t in texts))
        scored.append((with_text, len(members), len(comps), members))
    best = scored[0]
    for s in scored[1:]:                     # first among equal (with_text, size) pairs: a stable descending sort's head
        if (s[0], s[1]) > (best[0], best[1]):
            best = s
    if best[0] == 0 and best[2] > 0:
        main = clusters[0]
        for members in clusters[1:]:         # max(..., key=len): the first of the largest
            if len(members) > len(main):
                main = members
        plan["source"] = "main_cluster_fallback_no_text_assoc_in_best_with_components"
    else:
        main = best[3]
        plan["source"] = "main_yolo_cluster_scored_by_text_assoc"
    plan["main_size"] = len(main)
    bx0 = min(elems[i]["xmin"] for i in main)
    by0 = min(elems[i]["ymin"] for i in main)
    bx1 = max(elems[i]["xmax"] for i in main)
    by1 = max(elems[i]["ymax"] for i in main)
    plan["basis"] = (bx0, by0, bx1, by1)
    area = float(height * width)
    if area > 0 and (float(max(0, bx1 - bx0)) * float(max(0, by1 - by0))) / area > 0.90:
        plan["reason"] = "crop_basis_bbox_too_large"
        return plan
    x0, y0 = float(max(0, bx0 - padding)), float(max(0, by0 - padding))
    x1, y1 = float(min(width, bx1 + padding)), float(min(height, by1 + padding))
    plan["padded"] = (int(round(x0)), int(round(y0)), int(round(x1)), int(round(y1)))
    for t in texts:
        tx0, ty0, tx1, ty1 = float(t["xmin"]), float(t["ymin"]), float(t["xmax"]), float(t["ymax"])
        if tx1 < x0 - 150 or tx0 > x1 + 150 or ty1 < y0 - 150 or ty0 > y1 + 150:
            continue
        nx0, ny0 = min(x0, max(0, tx0 - 20)), min(y0, max(0, ty0 - 20))
        nx1, ny1 = max(x1, min(width, tx1 + 20)), max(y1, min(height, ty1 + 20))
        if (nx0, ny0, nx1, ny1) != (x0, y0, x1, y1):
            plan["text_uids"].append(t.get("persistent_uid"))
        x0, y0, x1, y1 = nx0, ny0, nx1, ny1
    wx0, wy0 = max(0, int(round(x0))), max(0, int(round(y0)))
    wx1, wy1 = min(width, int(round(x1))), min(height, int(round(y1)))
    plan["window"] = (wx0, wy0, wx1, wy1)
    if wx0 >= wx1 or wy0 >= wy1:
        plan["reason"] = "invalid_region_after_expansion"
        return plan
    plan["applied"] = True
    return plan


def shift_boxes(boxes, window):
    """:1256-1277: boxes in the window's coordinates, clipped to it; boxes left without positive area are dropped."""
    x0, y0, x1, y1 = window
    w, h = x1 - x0, y1 - y0
    out = []
    for b in boxes:
        nb = deepcopy(b)
        nb["xmin"], nb["ymin"] = max(0, b["xmin"] - x0), max(0, b["ymin"] - y0)
        nb["xmax"], nb["ymax"] = min(w, b["xmax"] - x0), min(h, b["ymax"] - y0)
        if nb["xmax"] > nb["xmin"] and nb["ymax"] > nb["ymin"]:
            out.append(nb)
    return out


def crop_image_and_adjust_bboxes(image, boxes, padding=20):
    """-> (cropped view or the image itself, adjusted boxes, plan)."""
    h, w = image.shape[:2]
    plan = crop_plan(boxes, h, w, padding)
    if not plan["applied"]:
        return image, [deepcopy(b) for b in boxes], plan
    x0, y0, x1, y1 = plan["window"]
    return image[y0:y1, x0:x1], shift_boxes(boxes, plan["window"]), plan
