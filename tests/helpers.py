"""Shared helpers for the GPU parity tests (HIP path vs the CPU oracle)."""
import torch

from circuitvision_amd import _lib
from circuitvision_amd._lib import F16, F32
from circuitvision_amd.engine import Buf, Plan, TORCH_DTYPE

TOL = {F16: dict(rtol=2e-2, atol=2e-2), F32: dict(rtol=1e-4, atol=1e-4)}


def stream():
    return torch.cuda.Stream()


def to_buf(x_nchw, dtype, c_total=None, c0=0):
    """NCHW float32 CPU tensor -> Buf (NHWC on device), optionally inside a wider buffer."""
    B, C, H, W = x_nchw.shape
    buf = Buf(B, H, W, c_total or C, dtype, zero=True)
    buf.t[..., c0:c0 + C] = x_nchw.permute(0, 2, 3, 1).to(TORCH_DTYPE[dtype]).cuda()
    return buf


def from_view(view):
    """View -> NCHW float32 CPU tensor."""
    return view.tensor().float().permute(0, 3, 1, 2).contiguous().cpu()


def quant(x, dtype):
    """Round a CPU f32 tensor to the storage dtype (so the oracle sees the same inputs)."""
    return x.to(TORCH_DTYPE[dtype]).float()


def run(plan):
    torch.cuda.synchronize()          # buffer fills ran on the default stream
    plan.run_eager()
    plan.stream.synchronize()
