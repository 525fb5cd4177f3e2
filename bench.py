#!/usr/bin/env python3
"""bench.py -- throughput of CircuitVision's dense-vision hot path on N MI355X (one process per GPU).

Default workload `circuit` = the metric BASELINE.json names, "circuit images/sec (YOLOv11 640^2 + SAM2.1-L 1024^2)":
a STEP pushes one batch of 32 circuit images per GPU through BOTH stages --
    YOLO11-n, 640x640, batch 32, fp16: forward + Detect decode + NMS           (BASELINE configs[1])
    SAM 2.1 Hiera-L, 1024x1024, 2 x batch 16, fp16 operands / f32 streams:
        image encoder + learned-prompt mask decoder + upsample / refinement     (BASELINE configs[2])
-- as three captured HIP graphs replayed back to back on one stream; inputs (letterboxed / normalised synthetic circuit
drawings) are resident in HBM before the timed region, weights are seeded random (nc = 62, LoRA merged).
`value` = images through both stages per second, whole job.  N > 1: every rank runs its own 32 images (weak scaling), the
weights are broadcast once from rank 0 over RCCL, there is no per-step collective.

Other workloads: yolo11n / yolo11l / sam2l / sam2l_box (one stage alone), pipeline (BASELINE configs[3]: YOLO11-l + SAM 2.1-L
over 64 images TOTAL, sharded over the ranks -- strong scaling).

`--gpus N` without a torchrun environment launches the N ranks itself (fresh child processes, before anything touches the
GPU); under torchrun WORLD_SIZE must equal --gpus.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant launch SHAPE of the workload, selected by MEASURED time share: launches of the profiling pass are grouped by
                (kernel the library dispatched to -- cvmi_last_kernel(), the name rocprofv3 lists --, flops, bytes) and the group with the largest
                summed duration wins (today: Hiera stage-3 fc2 = gemm256x192_kernel<float> or fc1 = tok_linear_kernel<576, 1, ...>);
                algorithmic flops per launch / average HIP-event duration of those launches; `traffic` = PMC bytes per launch of that kernel
                from the committed tools/traffic_sam.py passes.  `top_launches` lists the eight largest shapes the same way;
  rooflines     every family the north_star sets a target on: YOLO11-n conv stack vs the HBM roof (SURVEY.md 8(d) algorithmic
                bytes: 81.8 MB fp16 activations / image + 5.2 MB weights / batch), Hiera GEMMs, global and windowed attention vs
                the dense fp16 MFMA peak.  `achieved` = algorithmic bytes (flops) / the summed duration of those launches,
                measured here with HIP event pairs on the engine's stream in un-captured passes.
  cpu_baseline  the CPU fp32 oracle (oracle/) timed on this host: physical cores, CPU model and the sample are stated.
  stages        per-stage ms / images/s / launch breakdown.
"""
import argparse
import json
import math
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "circuit images/sec (YOLOv11 640² + SAM2.1-L 1024²) at 1/2/4/8 MI355X"
ALGO_ACT_MB_PER_IMAGE = 81.8     # 40.9 M fp16 activation elements moved / image
ALGO_WEIGHT_MB = 5.2             # 2.59 M params fp16, once per batch
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
SAM_FLOP_PER_IMAGE = 1.83e12     # trunk linear 1606 G + attention 203 G + conv/neck/decoder/refine
MFMA_PEAK_TFLOPS = 2500.0        # dense fp16/bf16 (MI355X_MICROARCH.md)
YOLO_STACK_KINDS = ("stem", "conv", "head", "dwconv", "pool", "attention")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="circuit", choices=["circuit", "yolo11n", "yolo11l", "sam2l", "sam2l_box", "pipeline"],
                    help="circuit = YOLO11-n B=32 + SAM2.1-L 2 x B=16 per step (default: the BASELINE metric); yolo11n = configs[1]; "
                         "sam2l = configs[2]; sam2l_box = one GPU's share of configs[4]; pipeline = configs[3] (64 images total, strong scaling)")
    ap.add_argument("--prompts", type=int, default=32, help="box prompts per image (sam2l_box)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per stage launch (default 32 YOLO / 16 SAM)")
    ap.add_argument("--total-images", type=int, default=64, help="pipeline workload: images in the whole job")
    ap.add_argument("--seg-batch", type=int, default=16, help="pipeline workload: images per segmenter launch (a rank's share is cut into such batches)")
    ap.add_argument("--scaling-proxy", type=int, default=0, metavar="N",
                    help="pipeline workload, 1 GPU: also measure ONE rank's share of an N-GPU job (total-images / N images) and report "
                         "`strong_scaling_proxy` = its images/s per GPU over this run's -- the only strong-scaling evidence a 1-GPU box can give")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32", "bf16"], help="operand storage type (bf16: SAM 2 workloads, BASELINE configs[4])")
    ap.add_argument("--attn", default="16", choices=["16", "fp8"], help="fp8: the AV products of Hiera's 256-key windows / global blocks on the block-scaled fp8 MFMA (BASELINE configs[4])")
    ap.add_argument("--streams", type=int, default=3, choices=[1, 2, 3, 4],
                    help="circuit / pipeline workloads: how the INDEPENDENT graphs of a step (detector batch, segmenter half-batches: different images) are "
                         "issued -- 1 = back to back on one stream; 2 = the segmenter batches alternate between two streams (detector with the first); "
                         "3 (default) = detector on its own stream as well; 4 = only the detector on its own stream.  Measured (profiles/r03_ab_runs.md): "
                         "two segmenter graphs side by side fill each other's kernel tails: +4.7 %% images/s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-cpu-baseline", action="store_true",
                    help="SAM 2.1-L CPU leg as BASELINE.md section 3 states it: the whole batch (B = 16), 1 warm-up + 5 timed iterations (~10 minutes of "
                         "host time; the default is a bounded batch-1 sample)")
    ap.add_argument("--no-profile-pass", action="store_true")
    a = ap.parse_args()
    sam_heavy = a.workload in ("circuit", "sam2l", "sam2l_box", "pipeline")
    if a.steps is None:
        a.steps = 5 if sam_heavy else 50
    if a.warmup is None:
        a.warmup = 2 if sam_heavy else 10
    return a


def self_launch(a):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child torchrun (nothing in THIS process has touched the
    GPU yet, and it never will) and exit with the child's code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


# ---- host description / CPU legs ---------------------------------------------------------------------------------------
def host_cpu():
    """(physical cores, model name) from /proc/cpuinfo."""
    cores, model, phys, core = set(), "unknown", None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))          # an affinity mask may expose fewer than the package has
    except AttributeError:
        pass
    try:                                                  # ... and so may a cgroup CPU quota ("<quota> <period>" or "max <period>")
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n), model


def yolo_cpu_baseline(scale, nc, state_dict, x, budget_s=40.0):
    """the CPU fp32 restatement on B = 32 synthetic 640x640 images, forward + NMS, 2 warm-up iterations,
    median of up to 20 timed iterations (bounded by `budget_s` of CPU time so that the default run stays within minutes)."""
    import torch
    from oracle import nms as onms
    from oracle.yolo11 import YOLO11
    cores, model = host_cpu()
    torch.set_num_threads(cores)
    m = YOLO11(scale, nc).eval()
    m.load_state_dict(state_dict, strict=True)
    ts = []
    with torch.no_grad():
        t_all = time.perf_counter()
        for i in range(22):
            t0 = time.perf_counter()
            onms.yolo_nms(m(x))
            dt = time.perf_counter() - t0
            if i >= 2:
                ts.append(dt)
            if time.perf_counter() - t_all > budget_s and len(ts) >= 3:
                break
    med = statistics.median(ts)
    return {"value": round(x.shape[0] / med, 3), "unit": "images/s", "cores": cores, "cpu_model": model, "kind": "port",
            "sample": f"YOLO11-{scale} fp32 oracle forward + NMS on the batch of {x.shape[0]} synthetic 640x640 images the GPU ran: 2 warm-up + "
                      f"{len(ts)} timed iterations (median {med:.3f} s, min {min(ts):.3f}, max {max(ts):.3f})"}


def sam_cpu_baseline(state_dict, x, boxes=None, timed=3, full=False):
    """BASELINE.md section 3 asks for B = 16 x >= 5 iterations; at several seconds per image on the host that is ~10 minutes, so the default
    sample is bounded: batch 1, 1 warm-up + `timed` iterations, median.  full=True (--full-cpu-baseline): the whole batch x 5 iterations."""
    import torch
    from oracle import sam2_model as osam
    cores, model = host_cpu()
    torch.set_num_threads(cores)
    w = osam.SAM2ImageWrapper(osam.SAM2Core(osam.HIERA_L, lora=True)).eval()
    w.load_state_dict({("sam2_model." + k if not (k.startswith("dense_") or k.startswith("sparse_") or k.startswith("refinement_")) else k): v
                       for k, v in state_dict.items()}, strict=True)
    ts = []
    nb = x.shape[0] if full else 1
    timed = 5 if full else timed
    with torch.no_grad():
        for i in range(1 + timed):
            t0 = time.perf_counter()
            if boxes is None:
                w(x[:nb])
            else:
                osam.predict_boxes(w, x[:nb], boxes[:nb])
            if i >= 1:
                ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    what = "wrapper forward (encoder + learned-prompt decoder + refinement)" if boxes is None else f"encoder + {boxes.shape[1]} box prompts"
    tail = ("BASELINE.md section 3's sample" if full else
            "bounded sample -- BASELINE.md's B=16 x 5 takes ~10 min on this host (--full-cpu-baseline runs it; one result in profiles/)")
    return {"value": round(nb / med, 4), "unit": "images/s", "cores": cores, "cpu_model": model, "kind": "port",
            "sample": f"SAM2.1 Hiera-L fp32 oracle {what}, batch {nb} of the synthetic 1024x1024 images the GPU ran: 1 warm-up + {len(ts)} timed "
                      f"iterations (median {med:.2f} s, min {min(ts):.2f}, max {max(ts):.2f}); {tail}"}


def synthetic_boxes(B, P, R=1024, seed=0):
    """P boxes per image, xyxy in the R x R input space, sides U(24, 200)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    side = 24 + 176 * torch.rand(B, P, 2, generator=g)
    xy = torch.rand(B, P, 2, generator=g) * (R - side)
    return torch.cat((xy, xy + side), -1)


def check_replicas(tensors, what):
    """After the weight broadcast every rank must hold the same bytes: a float64 checksum per rank, MIN and MAX over the ranks equal
    (non-zero ranks started from blank buffers, so equality also shows that the broadcast covers every tensor a forward pass reads)."""
    import torch
    import torch.distributed as dist
    if isinstance(tensors, dict):
        from circuitvision_amd.distributed import packed_tensors
        tensors = packed_tensors(tensors)
    s = torch.zeros(2, dtype=torch.float64, device="cuda")
    for i, t in enumerate(tensors):
        v = t.double()
        s[0] += v.sum() * (1 + (i % 7))
        s[1] += v.abs().sum()
    lo, hi = s.clone(), s.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi) or float(hi[1]) == 0.0:
        raise SystemExit(f"bench.py: {what} weight replicas differ between ranks after the broadcast ({lo.tolist()} vs {hi.tolist()})")


# ---- stages ------------------------------------------------------------------------------------------------------------------
class YoloStage:
    """YOLO11 forward + decode + NMS on B resident images (one captured graph)."""

    def __init__(self, a, scale, B, rank, local_rank, world, stream, seed0, lanes=None):
        import torch
        from circuitvision_amd import _lib
        from circuitvision_amd.distributed import broadcast_packed
        from circuitvision_amd.yolo11 import BlankParams, SyntheticParams, Yolo11Plan, Yolo11Weights
        from synth import circuit_image
        self.scale, self.B, self.nc = scale, B, 62
        if a.dtype == "bf16":
            raise SystemExit("--dtype bf16 is built for the SAM 2 path (workloads sam2l / sam2l_box); the detector runs fp16 / f32")
        self.dtype = {"f16": _lib.F16, "f32": _lib.F32}[a.dtype]
        # only rank 0 "reads the checkpoint" (generates + packs the seeded weights); the other ranks allocate blank buffers of the same
        # shapes and receive every packed tensor by the one-time RCCL broadcast
        self.params = SyntheticParams(seed=0, nc=self.nc) if rank == 0 else BlankParams()
        self.wt = Yolo11Weights(scale, self.nc, self.params, self.dtype, device=f"cuda:{local_rank}")
        if world > 1:
            broadcast_packed(self.wt.packed, src=0)            # one-time RCCL broadcast of the packed weights
            check_replicas(self.wt.packed, "YOLO11")
        self.yp = Yolo11Plan(self.wt, B, 640, 640, stream, keep_scores=False, lanes=lanes)
        lib = _lib.load()
        self.seeds = [seed0 + b for b in range(B)]
        for b, s in enumerate(self.seeds):                      # synthetic circuit images of this rank's shard, letterboxed on the GPU
            img = torch.from_numpy(circuit_image(640, 640, seed=s)).cuda()
            _lib.check(lib.cvmi_letterbox(img.data_ptr(), 640, 640, self.yp.x_in.t[b].data_ptr(), 640, 640, 640, 640, 0, 0, self.dtype, 1,
                                          stream.cuda_stream), "letterbox")
        stream.synchronize()
        self.plan = self.yp.plan
        # Seeded random weights never clear conf = 0.25, which would leave NMS without work.  All class logits of an anchor shift
        # together when the three class-conv biases do, so one shift (from one un-timed pass) puts the ~CAND highest-scoring anchors
        # per image above the threshold: the candidate volume of a busy schematic, sorted and suppressed inside the timed region.
        CAND = int(os.environ.get("CVMI_BENCH_CAND", "256"))      # (the environment knob is for NMS scaling experiments only)
        self.plan.run_eager()
        stream.synchronize()
        nc = self.nc
        lg = torch.cat([c.t[..., :nc].float().amax(-1).reshape(B, -1) for c in self.yp.cls_bufs], 1).flatten()
        kth = float(lg.kthvalue(lg.numel() - CAND * B + 1).values)
        for i in range(3):
            self.wt.packed[f"model.23.cv3.{i}.2"].bias[:nc] += (math.log(0.25 / 0.75) - kth + 1e-3)
            if rank == 0:
                self.params.sd[f"model.23.cv3.{i}.2.bias"] += (math.log(0.25 / 0.75) - kth + 1e-3)   # the CPU leg runs the same head
        self.plan.capture()

    def run(self):
        self.plan.run()

    def profile(self, reps=5):
        acc = {}
        self.plan.timed_eager()
        for _ in range(reps):
            for label, kind, ms, b, f in self.plan.timed_eager():
                k = acc.setdefault(kind, [0.0, 0, 0, 0])
                k[0] += ms / reps; k[1] += 1; k[2] += b / reps; k[3] += f / reps
        stack_ms = sum(acc[k][0] for k in YOLO_STACK_KINDS if k in acc)
        n_launch = sum(acc[k][1] for k in YOLO_STACK_KINDS if k in acc) // reps
        breakdown = {k: {"ms": round(v[0], 4), "launches": v[1] // reps, "plan_bytes": int(v[2]), "gflop": round(v[3] / 1e9, 3)} for k, v in acc.items()}
        if self.scale in ("n", "s"):
            is_cfg1 = self.scale == "n" and self.B == 32
            algo = (self.B * ALGO_ACT_MB_PER_IMAGE + ALGO_WEIGHT_MB) * 1e6 if is_cfg1 else \
                float(sum(acc[k][2] for k in YOLO_STACK_KINDS if k in acc)) + self.wt.param_bytes
            ach = algo / (stack_ms * 1e-3) / 1e9
            traffic, src = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if is_cfg1 and os.path.exists(tpath):                 # PMC pass of this exact workload (tools/traffic.py), not read in this run
                try:
                    t = json.load(open(tpath))
                    traffic, src = t.get("hbm_bytes_per_step"), f"profiles/traffic_latest.json ({t.get('collected', 'round 1, before the r02 kernels')}; " \
                        "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; a committed measurement, NOT read in this run)"
                except Exception:
                    pass
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": src,
                    "kernel": f"YOLO11-{self.scale} conv stack: AGGREGATE of its {n_launch} network launches per step (stem2 / c3k2 / conv_tile / igemm / dwpw / "
                              "dwconv / pool / attention), durations summed from per-launch event pairs; in the captured graph the Detect "
                              "lane overlaps the neck, so the step is shorter than this sum",
                    "kernel_ms_per_step": round(stack_ms, 4), "algorithmic_bytes_per_step": int(algo)}
        else:
            fl = float(sum(acc[k][3] for k in YOLO_STACK_KINDS if k in acc))
            ach = fl / (stack_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                    "traffic": None, "kernel": f"YOLO11-{self.scale} conv stack: aggregate of {n_launch} launches per step", "kernel_ms_per_step": round(stack_ms, 4),
                    "algorithmic_flops_per_step": int(fl)}
        return roof, breakdown

    def cpu_input(self):
        import numpy as np
        import torch
        from synth import circuit_image
        imgs = np.stack([circuit_image(640, 640, seed=s) for s in self.seeds])
        return torch.from_numpy(imgs[..., ::-1].copy()).permute(0, 3, 1, 2).float().div(255)


class SamStage:
    """SAM 2.1 Hiera-L on B resident images: learned-prompt wrapper forward (prompts = 0) or P box prompts per image."""

    _weights = {}

    def __init__(self, a, B, rank, local_rank, world, stream, seed0, prompts=0):
        import torch
        from circuitvision_amd import _lib
        from circuitvision_amd.distributed import broadcast_weights
        from circuitvision_amd.sam2 import HIERA_L, LORA_TARGETS_REFERENCE, Sam2Plan, Sam2Weights, SamBlankParams, SamSyntheticParams
        from synth import circuit_image
        self.B, self.P = B, prompts
        self.dtype = {"f16": _lib.F16, "f32": _lib.F32, "bf16": _lib.BF16}[a.dtype]
        key = (a.dtype, local_rank)
        if key not in SamStage._weights:                         # two SAM stages of one step share the weights
            params = SamSyntheticParams(seed=0, lora_targets=LORA_TARGETS_REFERENCE) if rank == 0 else SamBlankParams()
            wt = Sam2Weights(params, HIERA_L, 1024, self.dtype, device=f"cuda:{local_rank}")
            if world > 1:
                broadcast_weights(wt, src=0)                     # ranks > 0 hold blank buffers until this arrives
                from circuitvision_amd.distributed import weight_tensors
                check_replicas(weight_tensors(wt), "SAM 2.1")
            SamStage._weights[key] = (params, wt)
        self.params, self.wt = SamStage._weights[key]
        self.sp = Sam2Plan(self.wt, B, stream, prompts=prompts, attn=a.attn)
        self.boxes = None
        if prompts:
            self.boxes = synthetic_boxes(B, prompts, seed=rank)
            self.sp.coords[:, :2].copy_(self.boxes.reshape(B * prompts, 2, 2))
            self.sp.labels.copy_(torch.tensor([2, 3, -1], dtype=torch.int32).expand(B * prompts, 3))
        lib = _lib.load()
        self.seeds = [seed0 + b for b in range(B)]
        for b, s in enumerate(self.seeds):
            img = torch.from_numpy(circuit_image(768, 1024, seed=s)).cuda()
            _lib.check(lib.cvmi_sam2_transform(img.data_ptr(), 768, 1024, self.sp.x_in.t[b].data_ptr(), 1024, self.dtype, stream.cuda_stream), "transform")
        stream.synchronize()
        self.plan = self.sp.plan
        self.plan.capture()

    def run(self):
        self.plan.run()

    # launch kinds whose bound is the matrix pipe (flops / duration vs the dense MFMA peak); everything else is priced against HBM
    MFMA_KINDS = ("gemm", "mlp_fused", "attn_global", "attn_window", "attention")

    def _traffic(self):
        """Per-launch HBM bytes of the SAM pass from the committed PMC passes (tools/traffic_sam.py): {kernel: [entries]}."""
        path = os.path.join(ROOT, "profiles", "sam_traffic_latest.json")
        if self.P or self.B != 16 or self.dtype != 0 or not os.path.exists(path):      # collected for configs[2]: B = 16, fp16, learned prompts
            return {}, None
        try:
            t = json.load(open(path))
            return t.get("kernels", {}), f"profiles/sam_traffic_latest.json ({t.get('note', '')}; rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes over an eager pass; a committed measurement, NOT read in this run)"
        except Exception:
            return {}, None

    def profile(self, reps=2):
        """Per-launch event pairs over `reps` eager passes.  Launches are grouped into SHAPES = (kernel the library dispatched to, flops,
        plan bytes); the shape with the largest measured share of the pass is `roofline`, the next ones follow in `top_launches`."""
        acc, shapes = {}, {}
        self.plan.timed_eager()
        for _ in range(reps):
            for label, kind, ms, b, f, kn in self.plan.timed_eager(with_kernels=True):
                k = acc.setdefault(kind, [0.0, 0, 0, 0])
                k[0] += ms / reps; k[1] += 1; k[2] += b / reps; k[3] += f / reps
                if kind == "sync":
                    continue
                sh = shapes.setdefault((kn or f"[{kind}] {label.rsplit('.', 1)[-1]}", f, b), {"ms": 0.0, "n": 0, "kind": kind, "labels": []})
                sh["ms"] += ms; sh["n"] += 1
                if len(sh["labels"]) < 2 and label not in sh["labels"]:
                    sh["labels"].append(label)
        total_ms = sum(v[0] for v in acc.values())
        traffic, tsrc = self._traffic()
        ranked = sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])
        roofs, tops = {}, []
        for (kn, f, b), sh in ranked[:8]:
            n_pass = sh["n"] // reps
            us = sh["ms"] / sh["n"] * 1e3
            tr = next((e["hbm_bytes"] for e in traffic.get(kn, []) if e.get("launches_per_pass") == n_pass), None)
            # which roof: the one that gives the LARGER lower bound on the launch's time (the roofline model's ridge, MFMA peak / HBM peak =
            # 312 flop per byte): t_mfma = flops / dense peak, t_hbm = algorithmic (plan) bytes / HBM peak.  Both fractions are reported.
            t_mfma = f / (MFMA_PEAK_TFLOPS * 1e12) if sh["kind"] in self.MFMA_KINDS else 0.0
            t_hbm = b / (HBM_PEAK_GBS * 1e9)
            mfma = t_mfma >= t_hbm and f > 0
            ach = f / (us * 1e-6) / 1e12 if mfma else b / (us * 1e-6) / 1e9
            peak = MFMA_PEAK_TFLOPS if mfma else HBM_PEAK_GBS
            tops.append({"bound": "mfma" if mfma else "hbm", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s" if mfma else "GB/s",
                         "frac": round(ach / peak, 4), "traffic": tr, "traffic_source": tsrc if tr is not None else None,
                         "mfma_frac": round(t_mfma / (us * 1e-6), 4), "hbm_frac": round(t_hbm / (us * 1e-6), 4),
                         "kernel": f"{kn}: {n_pass} launches per B={self.B} pass (e.g. {', '.join(sh['labels'])}), {100 * sh['ms'] / reps / total_ms:.1f} % of the pass's summed "
                                   "launch time; average of HIP event pairs on the engine stream over un-captured passes",
                         "us_per_launch": round(us, 2), "launches_per_pass": n_pass, "share_of_pass": round(sh["ms"] / reps / total_ms, 4),
                         "algorithmic_flops_per_launch": int(f), "plan_bytes_per_launch": int(b),
                         "plan_bytes_gbs": round(b / (us * 1e-6) / 1e9, 1)})
        if tops:
            roofs["sam2l_dominant_launch"] = tops[0]
            roofs["sam2l_top_launches"] = tops
        if "mlp_fused" in acc:                                    # the fused stage-1 / 2 MLP launches are linear-layer flops too
            g, m = acc.setdefault("gemm", [0.0, 0, 0, 0]), acc["mlp_fused"]
            acc["linear"] = [g[0] + m[0], g[1] + m[1], g[2] + m[2], g[3] + m[3]]
        else:
            acc["linear"] = list(acc["gemm"])
        for name, kind, what in (("sam2l_linear_gemm", "linear", "Hiera / neck / decoder linear layers (tok_linear, hiera_mlp, gemm256*, igemm, gemm_glds kernels; LayerNorm and GELU fused where noted in DESIGN.md)"),
                                 ("sam2l_attention_global", "attn_global", "Hiera global attention, 3 blocks x 4096 x 4096 keys per head (attn_dma72_kernel)"),
                                 ("sam2l_attention_window", "attn_window", "Hiera windowed attention (attn_res256 / attn_res64 / attn_win16 kernels)")):
            if kind not in acc:
                continue
            ms, n, by, fl = acc[kind]
            t_mfma, t_hbm = fl / (MFMA_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9)          # the family's two lower bounds (ridge rule as above)
            mfma = t_mfma >= t_hbm
            ach = fl / (ms * 1e-3) / 1e12 if mfma else by / (ms * 1e-3) / 1e9
            peak = MFMA_PEAK_TFLOPS if mfma else HBM_PEAK_GBS
            roofs[name] = {"bound": "mfma" if mfma else "hbm", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s" if mfma else "GB/s",
                           "frac": round(ach / peak, 4), "mfma_frac": round(t_mfma / (ms * 1e-3), 4), "hbm_frac": round(t_hbm / (ms * 1e-3), 4),
                           "traffic": None, "kernel": f"{what}: aggregate of {n // reps} launches per B={self.B} pass",
                           "kernel_ms_per_step": round(ms, 3), "algorithmic_flops_per_step": int(fl), "plan_bytes_per_step": int(by)}
        roofs["sam2l_linear_gemm"]["whole_pass_tflops"] = round(self.B * SAM_FLOP_PER_IMAGE / (total_ms * 1e-3) / 1e12, 1)
        acc.pop("linear")
        breakdown = {k: {"ms": round(v[0], 3), "launches": v[1] // reps, "gflop": round(v[3] / 1e9, 1),
                         "tflops": round(v[3] / max(v[0], 1e-9) / 1e9, 1)} for k, v in acc.items()}
        return roofs, breakdown

    def cpu_input(self):
        import torch
        from oracle import sam2_model as osam
        from synth import circuit_image
        return torch.stack([osam.sam2_transform(circuit_image(768, 1024, seed=s), 1024) for s in (self.seeds if getattr(self, "full_cpu", False) else self.seeds[:1])])


# Label map of the host-inclusive pipeline: the crop decision (circuit_analyzer.py:937-1284) reads class names.  Component, junction, text and
# ignored classes in the proportions of a circuit label set.  NOTE: the seeded random detector of this bench returns ~25 boxes per image after the
# stage-2 NMS, all of ONE class (33) and each ~480 px wide (checked on the CPU oracle): the crop code runs (clustering, scoring, basis) and then
# declines -- "basis covers > 90 % of the page" -- so every window of this bench is the whole image.  The dependency (segmenter chunk k waits
# for detector chunk k's boxes on the host) and the glue's cost are real; non-trivial windows are exercised by tests/test_crop_gpu.py.
PIPE_NAMES = {i: (f"component{i}" if i % 10 in (0, 3) else "junction" if i % 10 == 1 else "text" if i % 10 == 2 else "explanatory") for i in range(62)}


class HostPipeline:
    """The pipeline as a caller runs it: `CircuitPipeline.run_batch` on u8 HOST images of this rank's shard, sharing the packed weights of
    the graph-only stages (analysis_pipeline.py:97-115 + :168-225 per image in the reference), WITH the reference's data dependency
    (crop=True): the segmenter's input is a window of the image that the detector's boxes decide (analysis_pipeline.py:177 -> :206).
    Inside the timed call: H2D, letterbox, detector, D2H of the detections, dicts + round() + uid + stage-2 NMS, crop window + box shift,
    channel swap + transform of the window, segmenter, post-process (resize to the window's size, threshold, u8, extent), D2H of the u8 masks.
    crop=False is the round-3 measurement (no dependency: segmenter and detector enqueued together), kept beside it."""

    def __init__(self, a, ystage, sstage, n, seed0, crop=True):
        import torch
        from circuitvision_amd.detector import YOLO
        from circuitvision_amd.pipeline import CircuitPipeline
        from circuitvision_amd.sam2 import HIERA_L
        from circuitvision_amd.sam2_infer import SAM2Model, SAM2Transforms
        from synth import circuit_image
        det = YOLO.from_weights(ystage.wt, PIPE_NAMES, dtype=a.dtype, graph_lanes=0)     # the stage's (bias-shifted) packed weights
        seg = SAM2Model(HIERA_L, 1024, dtype=a.dtype, use_refinement=True)
        seg.weights, seg.params = sstage.wt, sstage.params
        tr = SAM2Transforms(resolution=1024, mask_threshold=0, max_hole_area=0, max_sprinkle_area=0)
        self.pipe = CircuitPipeline(det, seg, tr, seg_batch=a.seg_batch, crop=crop, crop_padding=80)
        self.crop = crop
        self.images = [circuit_image(640, 640, seed=seed0 + i) for i in range(n)]
        self.n = n
        self.torch = torch

    def run(self):
        res = self.pipe.run_batch(self.images, "learned")
        t0 = time.perf_counter()
        masks = [r["mask"].cpu() for _, r in res]                       # the caller's `.detach().cpu()` (circuit_analyzer.py:355)
        self.pipe.timings["D2H of the u8 masks"] += time.perf_counter() - t0
        return res, masks

    def measure(self, steps, warmup):
        for _ in range(warmup):
            self.run()
        self.pipe.timings.clear()
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res, masks = self.run()
        self.torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        nb = sum(len(r["bboxes"]) for _, r in res)
        info = {"images_per_s": round(self.n / dt, 2), "ms_per_step": round(dt * 1e3, 3), "steps": steps,
                "mean_boxes_after_stage2_nms": round(nb / max(1, len(res)), 1),
                "phases_ms_per_step": {k: round(v / steps * 1e3, 3) for k, v in self.pipe.timings.items()},
                "what": "CircuitPipeline.run_batch(images u8 HxWx3 on the host, prompts='learned'" + (", crop=True, padding 80" if self.crop else "") +
                        ") + .cpu() of the u8 masks, wall clock around the calls"}
        if self.crop:
            wins = [r["window"] for _, r in res]
            fr = [((w[2] - w[0]) * (w[3] - w[1])) / float(self.images[i].shape[0] * self.images[i].shape[1]) for i, w in enumerate(wins) if w is not None]
            reasons = {}
            for _, r in res:
                k = r["crop_debug_info"]["reason_for_no_crop"] or "cropped"
                reasons[k] = reasons.get(k, 0) + 1
            info["crop"] = {"images_cropped": len(fr), "of": len(wins), "mean_window_area_fraction": round(sum(fr) / len(fr), 3) if fr else None,
                            "decisions": reasons,
                            "dependency": "segmenter chunk k is enqueued only after detector chunk k's boxes are on the host and its crop windows computed"}
        return info


# ---- main ---------------------------------------------------------------------------------------------------------------------
def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        self_launch(a)
    world = int(env_world or "1")
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torchrun --nproc-per-node {a.gpus} or drop the torchrun environment")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    proxy = None
    if a.scaling_proxy > 1 and a.workload == "pipeline" and world == 1:
        # one rank's share of an N-GPU run of the same job, measured in a child process BEFORE this one touches the GPU
        share = a.total_images // a.scaling_proxy
        if a.total_images % a.scaling_proxy or share % 2:
            raise SystemExit("--scaling-proxy N: total-images / N must be a whole, even number of images")
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", "pipeline", "--total-images", str(share), "--seg-batch", str(a.seg_batch),
               "--dtype", a.dtype, "--streams", str(a.streams), "--steps", str(a.steps * 4), "--warmup", str(a.warmup * 2), "--no-cpu-baseline", "--no-profile-pass"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f"bench.py: the --scaling-proxy child failed:\n{r.stderr[-2000:]}")
        proxy = json.loads(r.stdout.strip().splitlines()[-1])

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    stream = torch.cuda.Stream()
    w = a.workload
    scaling = "weak"
    host_pipe = None
    if w == "circuit":
        by, bs = a.batch or 32, 16
        nsam = by // bs
        if by % bs:
            raise SystemExit("--batch must be a multiple of 16 for the circuit workload")
        seed0 = 20250704 + rank * by
        # 1: one stream; 2: the two segmenter half-batches on two streams (detector with the first); 3: three streams; 4: detector on its own
        # stream, both segmenter half-batches on one
        extra = [torch.cuda.Stream() for _ in range(2)]
        sam_streams = [stream, extra[0]] if a.streams in (2, 3) else [stream, stream]
        # (a graph with internal side lanes on one of several concurrently used streams serialised the others in the r3m trace: the detector's
        #  graph is one linear chain whenever the step's graphs get streams of their own)
        stages = [("yolo11n", YoloStage(a, "n", by, rank, local_rank, world, extra[1] if a.streams in (3, 4) else stream, seed0,
                                        lanes=0 if a.streams > 1 else None))]
        stages += [(f"sam2l[{j}]", SamStage(a, bs, rank, local_rank, world, sam_streams[j % 2], seed0 + j * bs)) for j in range(nsam)]
        images_per_step = by
        name = (f"YOLO11-n 640x640 batch={by} {a.dtype} forward+decode+NMS (BASELINE configs[1]) + SAM2.1 Hiera-L 1024x1024 {nsam} x batch={bs} {a.dtype} "
                "learned-prompt wrapper forward (configs[2]) on the same images, per GPU and step")
    elif w in ("yolo11n", "yolo11l"):
        B = a.batch or 32
        stages = [(w, YoloStage(a, w[-1], B, rank, local_rank, world, stream, 20250704 + rank * B))]
        images_per_step = B
        name = f"YOLO11-{w[-1]} 640x640 batch={B}/GPU {a.dtype} forward+decode+NMS" + (" (BASELINE configs[1])" if w == "yolo11n" else "")
    elif w in ("sam2l", "sam2l_box"):
        B = a.batch or 16
        P = a.prompts if w == "sam2l_box" else 0
        stages = [(w, SamStage(a, B, rank, local_rank, world, stream, 20250704 + rank * B, prompts=P))]
        images_per_step = B
        attn = " + fp8 AV attention" if a.attn == "fp8" else ""
        name = (f"SAM2.1 Hiera-L 1024x1024 batch={B}/GPU {a.dtype}{attn} learned-prompt wrapper forward (BASELINE configs[2])" if not P else
                f"SAM2.1 Hiera-L 1024x1024 batch={B}/GPU, {P} box prompts per image, {a.dtype}{attn} (one GPU's share of BASELINE configs[4])")
    else:                                                        # pipeline: configs[3], strong scaling over a fixed total
        from circuitvision_amd.distributed import shard_range
        lo, hi = shard_range(a.total_images, rank, world)
        n = hi - lo
        if a.total_images % world or n % 2:
            raise SystemExit("--total-images must split evenly (and into an even share) over the ranks")
        bs = min(a.seg_batch, n)
        sam_sizes = [bs] * (n // bs) + ([n % bs] if n % bs else [])          # EVERY image of the shard goes through the segmenter
        assert sum(sam_sizes) == n
        seed0 = 20250704 + lo
        extra = [torch.cuda.Stream() for _ in range(2)]
        sam_streams = [stream, extra[0]] if a.streams in (2, 3) else [stream, stream]
        stages = [("yolo11l", YoloStage(a, "l", n, rank, local_rank, world, extra[1] if a.streams in (3, 4) else stream, seed0,
                                        lanes=0 if a.streams > 1 else None))]
        off = 0
        for j, b_ in enumerate(sam_sizes):
            stages.append((f"sam2l[{j}]", SamStage(a, b_, rank, local_rank, world, sam_streams[j % 2], seed0 + off)))
            off += b_
        images_per_step = n
        scaling = "strong"
        host_pipe = HostPipeline(a, stages[0][1], stages[1][1], n, seed0, crop=True)
        host_pipe_nocrop = HostPipeline(a, stages[0][1], stages[1][1], n, seed0, crop=False)
        name = (f"full pipeline YOLO11-l 640x640 + SAM2.1 Hiera-L 1024x1024, {a.total_images} circuit images per step sharded over {world} GPU(s) "
                f"({n} per GPU: detector batch {n}, segmenter batches {sam_sizes}), {a.dtype} (BASELINE configs[3]); `value` = graph replays on "
                "resident tensors, `host_inclusive` = CircuitPipeline.run_batch(crop=True) on u8 host images (H2D, letterbox, detector, D2H + glue + "
                "stage-2 NMS + crop window, transform of the window, segmenter, post-process, D2H of the u8 masks)")

    def run_step():
        for _, st in stages:
            st.run()

    for _ in range(a.warmup):
        run_step()
    stream.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run_step()
    stream.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
    ms_per_step = dt / a.steps * 1e3
    value = world * images_per_step * a.steps / dt
    host_info = None
    if host_pipe is not None:                                     # every rank runs its shard through the host-visible path too; MAX over ranks
        if world > 1:
            dist.barrier()
        host_info = host_pipe.measure(max(2, a.steps // 2), 1)
        if world > 1:
            t = torch.tensor([host_info["ms_per_step"]], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            host_info["ms_per_step"] = round(float(t.item()), 3)
            host_info["images_per_s"] = round(world * images_per_step / float(t.item()) * 1e3, 2)
        host_info["ratio_vs_graph_only"] = round(host_info["images_per_s"] / value, 4)
        nc_info = host_pipe_nocrop.measure(max(2, a.steps // 2), 1)
        host_info["without_crop_dependency"] = {"images_per_s": nc_info["images_per_s"], "ms_per_step": nc_info["ms_per_step"],
                                                "what": "the round-3 measurement: no crop, segmenter enqueued before the detector's boxes are on the host (rank-local)"}

    # ---- per-stage timing (graph replays alone) + per-launch timing pass (un-captured, event pairs on the engine stream)
    stage_info, rooflines, cpu_parts = {}, {}, {}
    if rank == 0:
        for sname, st in stages:
            if sname.endswith("]") and not sname.endswith("[0]"):
                continue                                           # identical second SAM stage
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3 if isinstance(st, SamStage) else 20
            sst = st.plan.stream                                   # (--streams > 1: a stage may own its stream)
            st.run(); sst.synchronize()
            e0.record(sst)
            for _ in range(reps):
                st.run()
            e1.record(sst); sst.synchronize()
            ms = e0.elapsed_time(e1) / reps
            key = sname.replace("[0]", "")
            info = {"batch": st.B, "ms_per_launch": round(ms, 4), "images_per_s": round(st.B / ms * 1e3, 2)}
            if not a.no_profile_pass:
                if isinstance(st, YoloStage):
                    roof, info["breakdown"] = st.profile()
                    rooflines[f"yolo11{st.scale}_conv_stack"] = roof
                    info["mean_detections_per_image"] = round(sum(st.yp.det_count.cpu().tolist()) / st.B, 1)
                else:
                    roofs, info["breakdown"] = st.profile()
                    rooflines.update(roofs)
            stage_info[key] = info
        if world == 1 and not a.no_cpu_baseline:
            for sname, st in stages:
                if sname.endswith("]") and not sname.endswith("[0]"):
                    continue
                key = sname.replace("[0]", "")
                if isinstance(st, YoloStage):
                    cpu_parts[key] = yolo_cpu_baseline(st.scale, st.nc, st.params.state_dict(), st.cpu_input())
                else:
                    st.full_cpu = a.full_cpu_baseline
                    cpu_parts[key] = sam_cpu_baseline(st.params.state_dict(), st.cpu_input(), boxes=st.boxes, full=a.full_cpu_baseline)

    if rank == 0:
        # the dominant kernel family by time: Hiera linear GEMMs wherever SAM runs, else the detector's conv stack
        # `roofline` = the single launch shape with the largest share of the step (its average duration can be checked against the rocprofv3
        # summary in profiles/); the family aggregates stay in `rooflines`
        tops = rooflines.pop("sam2l_top_launches", None)
        top = rooflines.get("sam2l_dominant_launch") or rooflines.get("sam2l_linear_gemm") or next(iter(rooflines.values()), None)
        cpu = None
        if cpu_parts:
            if len(cpu_parts) == 1:
                cpu = next(iter(cpu_parts.values()))
            else:                                                 # one image through both stages on the CPU: times add
                v = 1.0 / sum(1.0 / p["value"] for p in cpu_parts.values())
                first = next(iter(cpu_parts.values()))
                cpu = {"value": round(v, 4), "unit": "images/s", "cores": first["cores"], "cpu_model": first["cpu_model"], "kind": "port",
                       "sample": "one image through both stages = 1 / (1 / detector rate + 1 / segmenter rate); per-stage samples in `parts`",
                       "parts": cpu_parts}
        line = {
            "metric": METRIC, "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": name, "images_per_step": world * images_per_step, "nc": 62, "weights": "seeded random (SAM: LoRA merged)",
                       "streams": (a.streams if w in ("circuit", "pipeline") else 1)},
            "roofline": top, "cpu_baseline": cpu, "rooflines": rooflines, "top_launches": tops, "stages": stage_info,
        }
        if host_info is not None:
            line["host_inclusive"] = host_info
        if proxy is not None:
            hp, hi = proxy.get("host_inclusive", {}), host_info or {}
            line["strong_scaling_proxy"] = {
                "ranks_simulated": a.scaling_proxy, "images_per_rank": a.total_images // a.scaling_proxy,
                "graph_only": {"images_per_s_per_gpu_at_share": proxy["value"], "images_per_s_per_gpu_at_full": round(value, 3),
                               "ratio": round(proxy["value"] / value, 4)},
                "host_inclusive": {"images_per_s_per_gpu_at_share": hp.get("images_per_s"), "images_per_s_per_gpu_at_full": hi.get("images_per_s"),
                                   "ratio": round(hp["images_per_s"] / hi["images_per_s"], 4) if hp.get("images_per_s") and hi.get("images_per_s") else None},
                "what": f"ONE GPU running one rank's share ({a.total_images // a.scaling_proxy} images) of a {a.scaling_proxy}-GPU run of this {a.total_images}-image job, "
                        "against the same GPU running the whole job: images/s per GPU, share over full = the strong-scaling efficiency the sharded job "
                        "can reach when nothing else (host contention, RCCL start-up) is lost; measured on ONE GPU, no multi-GPU hardware curve exists",
                "share_run": {k: proxy.get(k) for k in ("value", "ms_per_step", "steps", "config")}, "share_host_inclusive": hp}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
