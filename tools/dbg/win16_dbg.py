import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch.nn.functional as F
from circuitvision_amd._lib import F16
from circuitvision_amd.engine import Plan, make_attn_desc, op_attention
from helpers import quant, run, stream
g = torch.Generator().manual_seed(13)
imgs, gh, gw, win, heads, hd = 1, 4, 8, 4, 1, 72
C_ = heads * hd
qkv = quant(torch.randn(imgs, gh, gw, 3 * C_, generator=g), F16)
def part(t):
    t = t.view(imgs, gh // win, win, gw // win, win, heads, hd).permute(0, 1, 3, 2, 4, 5, 6)
    return t.reshape(-1, win * win, heads, hd)
q, k, v = (part(qkv[..., i * C_:(i + 1) * C_]) for i in range(3))
scale = hd ** -0.5
s = torch.einsum("wqhd,wkhd->whqk", q, k) * scale
pr = torch.softmax(s, -1)
ref = torch.einsum("whqk,wkhd->wqhd", pr, v)
ref = ref.reshape(imgs, gh // win, gw // win, win, win, C_).permute(0, 1, 3, 2, 4, 5).reshape(imgs, gh, gw, C_)
qd = qkv.half().cuda().contiguous()
od = torch.zeros(imgs, gh, gw, C_, dtype=torch.float16, device="cuda")
nwin = imgs * (gh // win) * (gw // win)
desc = make_attn_desc(q=qd.data_ptr(), k=qd.data_ptr() + C_ * 2, v=qd.data_ptr() + 2 * C_ * 2, o=od.data_ptr(),
                      q_sb=0, q_sh=hd, q_st=3 * C_, k_sb=0, k_sh=hd, k_st=3 * C_, v_sb=0, v_sh=hd, v_st=3 * C_,
                      o_sb=0, o_sh=hd, o_st=C_, B=nwin, heads=heads, Nq=16, Nk=16, dqk=hd, dv=hd, scale=scale, dtype=F16,
                      win=win, grid_h=gh, grid_w=gw, q_pool=0)
plan = Plan(stream()); op_attention(plan, "t", desc, (qd, od)); run(plan)
got = od.float().cpu()
err = (got - ref).abs()
print("max err", err.max().item())
print("err per pixel (max over ch):"); print(err.amax(-1)[0])
print("got[0,0,0,:8]", got[0,0,0,:8], "ref", ref[0,0,0,:8])
# is got equal to V mean or some other?
print("v mean of window0 first 8:", v[0,:,0,:8].mean(0))
# --- uniform attention (K = 0): output must equal the window mean of V
qkv2 = qkv.clone(); qkv2[..., C_:2*C_] = 0
qd2 = qkv2.half().cuda().contiguous(); od.zero_()
desc.q = qd2.data_ptr(); desc.k = qd2.data_ptr() + C_*2; desc.v = qd2.data_ptr() + 2*C_*2
plan = Plan(stream()); op_attention(plan, "t", desc, (qd2, od)); run(plan)
got = od.float().cpu()
vm = part(qkv2[..., 2*C_:])[:, :, 0, :].mean(1)      # [nwin, hd]
print("uniform: got[0,0,0,:6]", got[0,0,0,:6], "window0 mean", vm[0,:6], "got[0,0,4,:6] (window1)", got[0,0,4,:6], "window1 mean", vm[1,:6])
# which key does it look like? compare got[0,0,0] to each V row of window 0
v0 = part(qkv2[..., 2*C_:])[0, :, 0, :]
print("dist to each key row:", [(round(float((got[0,0,0]-v0[k]).abs().max()),3)) for k in range(16)])
