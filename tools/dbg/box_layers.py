import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from circuitvision_amd import _lib
from circuitvision_amd.sam2 import HIERA_L, LORA_TARGETS_REFERENCE, Sam2Plan, Sam2Weights, SamSyntheticParams
wt = Sam2Weights(SamSyntheticParams(0, LORA_TARGETS_REFERENCE), HIERA_L, 1024, _lib.F16)
sp = Sam2Plan(wt, 16, torch.cuda.Stream(), prompts=32)
sp.x_in.t.normal_()
g = torch.Generator().manual_seed(0)
side = 24 + 176 * torch.rand(512, 2, generator=g); xy = torch.rand(512, 2, generator=g) * (1024 - side)
sp.coords[:, 0].copy_(xy); sp.coords[:, 1].copy_(xy + side)
sp.labels.copy_(torch.tensor([2, 3, -1], dtype=torch.int32).expand(512, 3))
torch.cuda.synchronize()
plan = sp.plan
plan.timed_eager()
acc = None
for _ in range(3):
    r = plan.timed_eager()
    acc = r if acc is None else [(l, k, m0 + m1, b, f) for (l, k, m0, b, f), (_, _, m1, _, _) in zip(acc, r)]
rows = [(l, k, ms / 3, b, f) for l, k, ms, b, f in acc if k in ("decoder", "tail", "neck")]
for l, k, ms, b, f in sorted(rows, key=lambda r: -r[2])[:40]:
    print(f"{l:28s} {k:8s} {ms*1e3:8.1f} us {b/1e6:9.1f} MB {b/ms/1e6 if ms>0 else 0:7.0f} GB/s {f/ms/1e9 if ms>0 else 0:7.1f} TF")
print("decoder+tail total ms", sum(r[2] for r in rows))
