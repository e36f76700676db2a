import sys, os, time; sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
from tests import gpu_checks as G
from signaltrain_amd.engine import StepEngine
from signaltrain_amd.dp import DataParallel
dist.init_process_group("nccl", rank=0, world_size=1)
B = 256
geo, X, Y, KN, P = G.make_case(B, 3, K=4)
d = G.dims_of(geo, B, 4)
eng = StepEngine(d, G.DEV); eng.load_state_dict(P)
x, k, y = (torch.from_numpy(a).to(G.DEV) for a in (X, KN, Y))
dp = DataParallel(eng, force_collectives=True, schedule="staged")
dp2 = DataParallel(eng, force_collectives=True, schedule="two_bucket")
def run(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
def stages_only():
    for s in range(4): eng.loss_backward_stage(s, x, k, y)
    eng.clip_adam(1e-4)
def stages_ar_sync():
    for s in range(4):
        eng.loss_backward_stage(s, x, k, y); dist.all_reduce(eng.stage_bucket(s))
    eng.clip_adam(1e-4)
def two_bucket():
    eng.loss_backward_p1(x, k, y); b = eng.grad_buckets()
    h0 = dist.all_reduce(b[0], async_op=True); eng.loss_backward_p2(); h1 = dist.all_reduce(b[1], async_op=True)
    h0.wait(); h1.wait(); eng.finish_buckets(); eng.clip_adam(1e-4)
def last_only():
    for s in range(4): eng.loss_backward_stage(s, x, k, y)
    dist.all_reduce(eng.stage_bucket(3)); eng.clip_adam(1e-4)
for name, fn in (("fused train_step", lambda: eng.train_step(x, k, y, 1e-4)), ("4 stages, no collectives", stages_only),
                 ("DataParallel staged (4 stages + 4 async all-reduce)", lambda: dp.train_step(x, k, y, 1e-4)), ("DataParallel two_bucket (default)", lambda: dp2.train_step(x, k, y, 1e-4)), ("4 stages + 4 in-stream all-reduce", stages_ar_sync),
                 ("p1/p2 + 2 async all-reduce (old)", two_bucket), ("4 stages + 1 all-reduce", last_only)):
    h, t = run(fn); print(f"{name:52s} host issue {h:.3f} ms/step   total {t:.3f} ms/step", flush=True)
