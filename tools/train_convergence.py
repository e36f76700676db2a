#!/usr/bin/env python3
"""End-to-end accuracy of the arithmetic modes: the SAME training run (same initial weights, same device-generated comp_4c
minibatches, same 1-cycle schedule) in every compute_dtype, compared by the training loss averaged over windows of steps and by
the loss on a fixed validation batch.  The reference's own loop is what is imitated (train.py:104-151: forward, loss, backward,
L1 clip, Adam, lr write), driven through StepEngine.train_step so that nothing but the arithmetic differs between the runs.
    python tools/train_convergence.py [steps] [batch]        (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from signaltrain_amd import _lib, nn_proc, audio, datasets, learningrate
from signaltrain_amd.engine import StepEngine
nn_proc._QUIET = True
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
MODES = ("f32", "f32+1e-6", "f32+seed", "f32x3", "bf16", "bf16_all", "f16", "f16_all",
         "f32/ca1", "bf16_all/ca1", "f16_all/ca0")     # f32+1e-6: fp32 from initial weights perturbed by 1e-6 relative; f32+seed: fp32 with the minibatch order reversed -- the
                                                        # spread of the loss itself; /ca1, /ca0: the L1 clip over ALL parameters (train.py:136, the f16 modes' default) switched on / off:
                                                        # separates the clip SCOPE from the operand ROUNDING as the cause of the 16-bit modes' lower final loss
if len(sys.argv) > 3:
    MODES = tuple(sys.argv[3].split(","))

torch.manual_seed(218); np.random.seed(218)
model = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=4)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
ds = datasets.SynthAudioDataSet(8192, audio.Compressor_4c(), datapoints=STEPS * B, y_size=2048)
gen = torch.Generator(device=dev); gen.manual_seed(218)
CH = 32                                                   # minibatches generated per device call
data = []
for c in range((STEPS + CH - 1) // CH):
    x, y, kn = ds.batch_device(CH * B, dev, generator=gen)
    data.append((x, y, kn))
xv, yv, kv = ds.batch_device(256, dev, generator=gen)     # fixed validation batch
lrs, _ = learningrate.get_1cycle_schedule(lr_max=1e-3, n_data_points=STEPS * B, epochs=1, batch_size=B)
d = _lib.geometry(1, 4, 4, B); dv = _lib.geometry(1, 4, 4, 256)

def val_loss(eng_params):
    ev = StepEngine(dv, dev); ev.params.copy_(eng_params)          # always evaluated in fp32
    ev.loss_backward(xv, kv, yv); torch.cuda.synchronize()
    return float(ev.scalars[0])

rows = {}
for mode in MODES:
    base = mode.split("+")[0].split("/")[0]
    ca = {"ca1": True, "ca0": False}.get(mode.split("/")[1]) if "/" in mode else None
    eng = StepEngine(d, dev, compute_dtype=base, clip_all=ca); eng.load_state_dict(sd)
    if mode == "f32+1e-6":
        g = torch.Generator(device=dev); g.manual_seed(1)
        eng.params.mul_(1.0 + 1e-6 * torch.randn(eng.params.shape, device=dev, generator=g))
    losses = torch.zeros(STEPS, device=dev)
    for it in range(STEPS):
        jt = STEPS - 1 - it if mode == "f32+seed" else it
        x, y, kn = data[jt // CH]; sl = slice((jt % CH) * B, (jt % CH + 1) * B)
        sc = eng.train_step(x[sl], kn[sl], y[sl], float(lrs[max(it - 1, 0)]))
        losses[it] = sc[0]
    torch.cuda.synchronize()
    l = losses.cpu().numpy()
    q = STEPS // 4
    rows[mode] = dict(first=float(l[:10].mean()), quarters=[float(l[i * q:(i + 1) * q].mean()) for i in range(4)], last50=float(l[-50:].mean()),
                      val=val_loss(eng.params), skipped=int(eng.scalars[5]))
ref = rows["f32"]
print(f"{STEPS} steps of batch {B} (comp_4c windows generated on the device, 1-cycle lr to 1e-3), training loss = mean over the steps of each quarter; validation: 256 fixed windows, fp32 forward")
print(f"{'mode':12s} {'first 10':>10s} {'Q1':>10s} {'Q2':>10s} {'Q3':>10s} {'Q4':>10s} {'last 50':>10s} {'validation':>11s} {'val vs f32':>10s} {'skipped steps':>14s}")
for mode, r in rows.items():
    print(f"{mode:12s} {r['first']:10.3e} " + " ".join(f"{v:10.3e}" for v in r["quarters"]) + f" {r['last50']:10.3e} {r['val']:11.3e} {r['val'] / ref['val'] - 1:+10.1%} {r['skipped']:14d}")
