import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from circuitvision_amd import _lib
from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Plan, Yolo11Weights
wt = Yolo11Weights("n", 62, SyntheticParams(0, 62), _lib.F16)
def run(parts, B=32, steps=50):
    plans = []
    for i in range(parts):
        st = torch.cuda.Stream()
        yp = Yolo11Plan(wt, B // parts, 640, 640, st)
        yp.set_input_nchw(torch.rand(B // parts, 3, 640, 640))
        plans.append(yp)
    torch.cuda.synchronize()
    for yp in plans: yp.plan.capture()
    for _ in range(5):
        for yp in plans: yp.plan.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for yp in plans: yp.plan.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"parts={parts}: {dt*1e3:.3f} ms per {B} images -> {B/dt:.0f} img/s", flush=True)
for parts in (1, 2, 4, 1, 2):
    run(parts)
