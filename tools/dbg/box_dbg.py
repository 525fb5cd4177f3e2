import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_oracle_sam2_cpu import MINI, mini_oracle, mini_targets
from oracle import sam2_model as osam
from circuitvision_amd.sam2 import Sam2Plan, Sam2Weights, SamSyntheticParams
from circuitvision_amd._lib import F32
from test_sam2_gpu import _boxes

R = 256
for seed_p, B, P in ((5, 2, 5), (8, 2, 3), (8, 2, 5), (5, 2, 3)):
    p = SamSyntheticParams(seed=seed_p, lora_targets=mini_targets(), std=0.05)
    wt = Sam2Weights(p, MINI, R, F32)
    oracle = mini_oracle(p, R)
    x = torch.randn(B, 3, R, R, generator=torch.Generator().manual_seed(2))
    sp = Sam2Plan(wt, B, torch.cuda.Stream(), prompts=P)
    for bs in (1, 2):
        boxes = _boxes(B, P, R, seed=bs)
        with torch.no_grad():
            rhi, rlo, riou = osam.predict_boxes(oracle, x, boxes)
        sp.x_in.t.copy_(x.permute(0, 2, 3, 1))
        sp.coords[:, :2].copy_(boxes.reshape(B * P, 2, 2))
        sp.labels.copy_(torch.tensor([2, 3, -1], dtype=torch.int32).expand(B * P, 3))
        torch.cuda.synchronize()
        for mode in ("eager", "graph"):
            sp.low_res.zero_()
            torch.cuda.synchronize()
            (sp.plan.run_eager if mode == "eager" else sp.plan.run)()
            torch.cuda.synchronize()
            e = (sp.low_res.view(B, P, R // 4, R // 4).cpu() - rlo).abs().amax((2, 3))
            print(seed_p, B, P, "boxes", bs, mode, "err per prompt", e.flatten().tolist(), "sel", sp.sel.cpu().tolist(), flush=True)
