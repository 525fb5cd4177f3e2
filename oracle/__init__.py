"""CPU fp32 restatement of the reference's dense-vision hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (circuitvision_amd/) never does: it fails loudly when the HIP library is missing.

Parity status: the arithmetic of this path lives in un-vendored, un-pinned third-party packages
(ultralytics, facebookresearch/sam2, peft, torchvision: /root/reference/requirements.txt:7-13)
whose sources are not under /root/reference, and the reference holds no tests.  Pinned parts:
  * oracle/nms.py stage-2 NMS           <- golden vectors from src/utils.py:297-361
  * oracle/sam2_tail.py refinement head <- golden vectors from src/sam2_infer.py:130-189
  * oracle/sam2_tail.py postprocess     <- golden vectors from src/sam2_infer.py:88-128
  * oracle/sam2_*.py trunk/neck/decoder <- cross-checked against the independent transformers
                                           SAM2 implementation (CPU test, build container only)
Everything else (YOLO11 graph, ultralytics NMS, letterbox, torchvision resize) is "parity
unpinned": restated from the published upstream algorithm and anchored on known-answer
parameter/FLOP counts.
"""
