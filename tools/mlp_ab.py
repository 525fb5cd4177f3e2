"""A/B of hiera_mlp variants (profiles/r04_ab_runs.md): one process per variant (the env switches are read once). Prints a checksum of the output and the time per launch."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from circuitvision_amd._lib import BF16, F16, F32
from circuitvision_amd.engine import Buf, Plan, PackedHieraMlp, op_hiera_mlp
from tests.helpers import stream, run
C_ = int(sys.argv[1]); rows = int(sys.argv[2]); dt = BF16 if len(sys.argv) > 3 and sys.argv[3] == "bf16" else F16
g = torch.Generator().manual_seed(1)
x = torch.randn(rows, C_, generator=g) * 1.5 + 0.3
gam, bet = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.2
w1 = torch.randn(4 * C_, C_, generator=g) / C_ ** 0.5
b1 = torch.randn(4 * C_, generator=g) * 0.3
w2 = torch.randn(C_, 4 * C_, generator=g) / (4 * C_) ** 0.5
b2 = torch.randn(C_, generator=g) * 0.3
pm = PackedHieraMlp(w1, b1, w2, b2, dtype=dt)
xb = Buf(1, 1, rows, C_, F32)
xb.t.copy_(x.view(1, 1, rows, C_))
plan = Plan(stream())
stats = torch.zeros((rows, 2), device="cuda")
op_hiera_mlp(plan, "mlp", pm, xb.images(0, 1).view(), gam.cuda(), bet.cuda(), 1e-6, stats_out=stats, stats_eps=1e-6)
run(plan)
torch.cuda.synchronize()
out = xb.t.cpu().numpy().tobytes() + stats.cpu().numpy().tobytes()
h = hashlib.sha256(out).hexdigest()[:16]
import torch.nn.functional as TF
from circuitvision_amd.engine import TORCH_DTYPE
td = TORCH_DTYPE[dt]
q = lambda t: t.to(td).float()
xc = x.cuda()
xn_ = q(TF.layer_norm(xc, (C_,), gam.cuda(), bet.cuda(), 1e-6))
hid = q(TF.gelu(xn_ @ q(w1).cuda().t() + b1.cuda()))
ref = xc + hid @ q(w2).cuda().t() + b2.cuda()
got = xb.t.view(rows, C_)
err = (got - ref).abs()
tag = f"{C_}_{rows}_{'bf16' if dt == BF16 else 'f16'}"
od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "mlp_ab")
os.makedirs(od, exist_ok=True)
key = os.environ.get("CVMI_MLP_PIPE", "0")
other = os.path.join(od, f"{tag}_pipe0.pt")
cmp = ""
if key == "0": torch.save(got[:4096].cpu(), other)
elif os.path.exists(other):
    o = torch.load(other)
    d = (got[:4096].cpu() - o).abs()
    cmp = f" vs pipe0 (first 4096 rows): max diff {d.max().item():.3e}, differing {int((d > 0).sum())} of {d.numel()}"
print(f"   err vs torch: max {err.max().item():.3e} mean {err.mean().item():.3e}{cmp}")
s = torch.cuda.Stream()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(3): run(plan)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
N = 20
for _ in range(N): plan.run_eager()
plan.stream.synchronize()
dt_us = (time.perf_counter() - t0) / N * 1e6
print(f"C={C_} rows={rows} {'bf16' if dt == BF16 else 'f16'} PIPE={os.environ.get('CVMI_MLP_PIPE','-')} VAR={os.environ.get('CVMI_MLP_VAR','-')}: sha {h}  {dt_us:.1f} us/launch  finite={bool(torch.isfinite(xb.t).all())}")
