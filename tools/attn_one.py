#!/usr/bin/env python3
"""Run ONE Hiera attention site a few times (for rocprofv3 --pmc / --kernel-trace): stage-3 windows or global."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from circuitvision_amd._lib import F16
from circuitvision_amd.engine import Plan, make_attn_desc, op_attention
mode = sys.argv[1] if len(sys.argv) > 1 else "win16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
imgs, g, heads, hd = 16, 64, 8, 72
C_ = heads * hd
qkv = torch.randn(imgs, g, g, 3 * C_, device="cuda", dtype=torch.float16)
o = torch.zeros(imgs, g, g, C_, device="cuda", dtype=torch.float16)
if mode == "win16":
    win = 16; nwin = imgs * (g // win) ** 2
    d = make_attn_desc(q=qkv.data_ptr(), k=qkv.data_ptr() + C_ * 2, v=qkv.data_ptr() + 2 * C_ * 2, o=o.data_ptr(), q_sb=0, q_sh=hd, q_st=3 * C_,
                       k_sb=0, k_sh=hd, k_st=3 * C_, v_sb=0, v_sh=hd, v_st=3 * C_, o_sb=0, o_sh=hd, o_st=C_, B=nwin, heads=heads, Nq=256, Nk=256,
                       dqk=hd, dv=hd, scale=hd ** -0.5, dtype=F16, win=win, grid_h=g, grid_w=g, q_pool=0)
else:
    N = g * g
    d = make_attn_desc(q=qkv.data_ptr(), k=qkv.data_ptr() + C_ * 2, v=qkv.data_ptr() + 2 * C_ * 2, o=o.data_ptr(), q_sb=N * 3 * C_, q_sh=hd, q_st=3 * C_,
                       k_sb=N * 3 * C_, k_sh=hd, k_st=3 * C_, v_sb=N * 3 * C_, v_sh=hd, v_st=3 * C_, o_sb=N * C_, o_sh=hd, o_st=C_, B=imgs, heads=heads,
                       Nq=N, Nk=N, dqk=hd, dv=hd, scale=hd ** -0.5, dtype=F16, win=0, grid_h=0, grid_w=0, q_pool=0)
st = torch.cuda.Stream()
plan = Plan(st)
op_attention(plan, "a", d, (qkv, o))
torch.cuda.synchronize()
plan.run_eager(); st.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(reps):
    plan.run_eager()
e1.record(st); st.synchronize()
print(mode, "us per launch", e0.elapsed_time(e1) * 1e3 / reps)
