#!/usr/bin/env python3
"""Round-2 golden vectors from the *imported reference* (build container only); complements tools/capture_golden.py.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_r2.py
Writes (small, committed) and asserts oracle/st_oracle.py against every value:

  tests/golden/g4b_backward_clip.npz   one backward with an ACTIVE L1 clip (the reference's clip_grad_norm_ returns n > 1,
                                        nn_proc.py:299-302): loss, all gradients (fingerprinted), clip norm / coefficient,
                                        the clipped STFT gradients
  tests/golden/g5b_adam_clip.npz       three optimisation steps in the order of train.py:131-151 with the clip active on
                                        every step (parameters after each step)
  tests/golden/g10_checkpoint.npz      HEADER of a checkpoint written by the reference's misc.save_checkpoint
                                        (misc.py:21-35) after one optimisation step with torch.optim.Adam: top-level keys,
                                        state_dict keys / shapes / dtypes, the optimizer state_dict structure, sampled tensors.
                                        The 50 MB file itself is not committed; this script also checks, here, that
                                        signaltrain_amd.misc.load_checkpoint + st_model.load_state_dict + the engine's
                                        optimizer restore read the real file.
G4/G5 of the first capture run with clip_coef = 1.0 (norm 0.216), so the n > 1 branch was pinned only through the oracle.
"""
import os, sys, tempfile
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _ref_import import import_reference                       # noqa: E402
from oracle import st_oracle as O                                # noqa: E402
from tests.golden_util import perturb_stft, projections, ae_keys, SAMPLE_ROWS  # noqa: E402

R = import_reference()
nn_proc, loss_functions, learningrate, ref_misc, ref_audio = R.nn_proc, R.loss_functions, R.learningrate, R.misc, R.audio
torch.set_num_threads(8)


def report(name, a, b, tol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.max(np.abs(a - b)) if a.size else 0.0
    scale = max(np.max(np.abs(b)), 1e-30) if b.size else 1.0
    print(f"  {name:34s} max|d|={err:.3e} rel={err/scale:.3e}")
    assert err <= tol * max(scale, 1e-30) + 1e-30, f"oracle mismatch on {name}: {err} vs scale {scale}"


# ---- parameters: AE weights of the committed G3 fixture + regenerated, perturbed ("learned") STFT bases -------------
geo = O.geometry(1, 4)
g3 = np.load(os.path.join(OUT, "g3_forward.npz"))
P = O.init_params(geo, 4)
for k in ae_keys():
    P[k] = g3["ae_" + k]
perturb_stft(P, seed=7)
model = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=4)
with torch.no_grad():
    model.load_state_dict({k: torch.from_numpy(P[k]) for k in P})
model.train()

# ---- inputs: three comp_4c windows whose TARGETS carry a gain error (x 1.6) and a sign flip on one window, i.e. a model far
# from its target -- that is what the first steps of a real run look like ("always active in practice") ----
rng = np.random.default_rng(2182)
B = 3
X, Y, KN = O.synth_comp4c_batch(B, geo["L"], geo["y"], rng)
Y = (Y * np.float32(1.6)).astype(np.float32)
Y[1] = -Y[1]
xt, kt, yt = torch.from_numpy(X), torch.from_numpy(KN), torch.from_numpy(Y)
F = geo["F"]

# ---- G4b ---------------------------------------------------------------------------------------------------------------
print("G4b backward, active clip")
y_hat, mag, mag_hat = model.forward(xt, kt)
sbf = torch.exp((7. / F) * torch.arange(0., F)).expand_as(mag_hat).float()
loss = loss_functions.calc_loss(y_hat.float(), yt.float(), mag_hat.float(), scale_by_freq=sbf)
model.zero_grad(); loss.backward()
gref = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
ol32, og32, _ = O.model_loss_bwd(X, KN, Y, P, geo)
ol64, og64, _ = O.model_loss_bwd(X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64), P, geo)
report("loss (f32 oracle)", ol32, loss.item(), 3e-5)
for k in gref:
    report("grad " + k.replace("mpaec.", "")[:28], og64[k], gref[k], 2e-4)
total_norm = torch.nn.utils.clip_grad_norm_(
    list(model.mpaec.dft_analysis.parameters()) + list(model.mpaec.dft_synthesis.parameters()), max_norm=1., norm_type=1)
n_ref = float(total_norm.item())
coef_ref = min(1.0, 1.0 / (n_ref + 1e-6))
print(f"   reference clip norm {n_ref:.6f} -> coef {coef_ref:.6f}")
assert n_ref > 1.5, "the clip must be active for this fixture"
gclip = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}       # STFT grads now scaled in place
og_c = {k: v.copy() for k, v in og32.items()}
n_o, c_o = O.clip_l1_stft(og_c)
report("clip norm", n_o, n_ref, 1e-3)            # sum|g| over 4 M fp32 values: order-sensitive at ~2e-4
report("clip coef", c_o, coef_ref, 1e-3)
PROJ = projections(seed=11)
g4b = dict(x=X, knobs=KN, y=Y, loss=np.float64(loss.item()), clip_norm=np.float64(n_ref), clip_coef=np.float64(coef_ref))
for k in ae_keys():
    g4b["g_" + k] = gref[k]
for k in O.STFT_KEYS:
    g = gref[k][:, 0, :]; gc = gclip[k][:, 0, :]
    report("clipped " + k.split(".")[-2], og_c[k][:, 0, :], gc, 1e-3)
    g4b["l1_" + k] = np.float64(np.abs(g.astype(np.float64)).sum())
    g4b["rows_" + k] = g[SAMPLE_ROWS]
    g4b["proj_" + k] = PROJ @ g.astype(np.float64)
    g4b["clipped_rows_" + k] = gc[SAMPLE_ROWS]
    g4b["clipped_proj_" + k] = PROJ @ gc.astype(np.float64)
    g4b["clipped_l1_" + k] = np.float64(np.abs(gc.astype(np.float64)).sum())

# ---- G5b: three steps, train.py:131-151 ordering, clip active on each --------------------------------------------------
print("G5b adam x3, active clip")
with torch.no_grad():
    model.load_state_dict({k: torch.from_numpy(P[k]) for k in P})
lrs, _ = learningrate.get_1cycle_schedule(lr_max=1e-3, n_data_points=300, epochs=1, batch_size=B)
opt = torch.optim.Adam(list(model.parameters()), lr=lrs[0], weight_decay=0)
Pq = {k: P[k].copy() for k in O.param_order()}
Mq = {k: np.zeros_like(v) for k, v in Pq.items()}
Vq = {k: np.zeros_like(v) for k, v in Pq.items()}
g5b = dict(lrs=lrs[:4].copy())
for it in range(3):
    Xi = np.roll(X, 23 * it, axis=1).copy(); Yi = np.roll(Y, 23 * it, axis=1).copy()
    lr_used = opt.param_groups[0]["lr"]
    yh, mg, mh = model.forward(torch.from_numpy(Xi), kt)
    ls = loss_functions.calc_loss(yh.float(), torch.from_numpy(Yi).float(), mh.float(), scale_by_freq=sbf)
    opt.zero_grad(); ls.backward()
    nrm = torch.nn.utils.clip_grad_norm_(
        list(model.mpaec.dft_analysis.parameters()) + list(model.mpaec.dft_synthesis.parameters()), max_norm=1., norm_type=1)
    # == model.clip_grad_norm_() (nn_proc.py:299-302), called directly to record the norm it acts on
    opt.step()
    opt.param_groups[0]["lr"] = lrs[it]                      # train.py:150
    lo, no, co = O.train_step(Xi, KN, Yi, Pq, Mq, Vq, it + 1, lr_used, geo)
    assert nrm.item() > 1.2, ("clip inactive at step", it, nrm.item())
    report(f"step{it} loss", lo, ls.item(), 3e-5)
    report(f"step{it} clip norm", no, nrm.item(), 1e-3)
    sdn = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    g5b[f"loss{it}"] = np.float64(ls.item()); g5b[f"lr_used{it}"] = np.float64(lr_used); g5b[f"clip_norm{it}"] = np.float64(nrm.item())
    worst = max(np.max(np.abs(Pq[k].astype(np.float64) - sdn[k])) for k in O.param_order())
    print(f"  step{it} max |param diff| oracle vs ref = {worst:.3e}")
    assert worst < 5e-6
    for k in ae_keys():
        g5b[f"s{it}_" + k] = sdn[k]
    for k in O.STFT_KEYS:
        g5b[f"s{it}_rows_" + k] = sdn[k][SAMPLE_ROWS, 0, :]
        g5b[f"s{it}_proj_" + k] = PROJ @ sdn[k][:, 0, :].astype(np.float64)

# ---- G10: header of a reference-written checkpoint -----------------------------------------------------------------------
print("G10 checkpoint written by the reference's misc.save_checkpoint")
eff = ref_audio.Compressor_4c()
opt.param_groups[0]["momentum"] = 0.9123                      # train.py:151 writes this (ignored) key every iteration
tmp = tempfile.mkdtemp(prefix="st_ckpt_")
ck = os.path.join(tmp, "modelcheckpoint.tar")
ref_misc.save_checkpoint(ck, model, 41, False, opt, eff, 44100)
print(f"   ({os.path.getsize(ck)/1e6:.1f} MB, not committed)")
raw = torch.load(ck, map_location="cpu", weights_only=False)
osd = raw["optimizer"]
g10 = dict(
    top_keys=np.array(list(raw.keys())),
    epoch=np.int64(raw["epoch"]), effect_name=np.array(raw["effect_name"]), knob_names=np.array(raw["knob_names"]),
    knob_ranges=np.asarray(raw["knob_ranges"]), knob_ranges_dtype=np.array(str(np.asarray(raw["knob_ranges"]).dtype)),
    scale_factor=np.int64(raw["scale_factor"]), shrink_factor=np.int64(raw["shrink_factor"]),
    in_chunk_size=np.int64(raw["in_chunk_size"]), out_chunk_size=np.int64(raw["out_chunk_size"]), sr=np.int64(raw["sr"]),
    sd_keys=np.array(list(raw["state_dict"].keys())),
    sd_shapes=np.array([",".join(map(str, v.shape)) for v in raw["state_dict"].values()]),
    sd_dtypes=np.array([str(v.dtype) for v in raw["state_dict"].values()]),
    opt_top_keys=np.array(list(osd.keys())),
    opt_state_ids=np.array(list(osd["state"].keys()), np.int64),
    opt_state_keys=np.array(list(osd["state"][0].keys())),
    opt_state_shapes=np.array([",".join(map(str, osd["state"][i]["exp_avg"].shape)) for i in osd["state"]]),
    opt_step_is_tensor=np.bool_(torch.is_tensor(osd["state"][0]["step"])),
    opt_step=np.float64(float(osd["state"][0]["step"])),
    opt_group_keys=np.array(sorted(osd["param_groups"][0].keys())),
    opt_group_params=np.array(osd["param_groups"][0]["params"], np.int64),
    opt_lr=np.float64(osd["param_groups"][0]["lr"]), opt_betas=np.array(osd["param_groups"][0]["betas"], np.float64),
    opt_eps=np.float64(osd["param_groups"][0]["eps"]),
)
names = list(raw["state_dict"].keys())
for i, k in enumerate(names):
    w = raw["state_dict"][k].numpy(); m_, v_ = osd["state"][i]["exp_avg"].numpy(), osd["state"][i]["exp_avg_sq"].numpy()
    if k in O.STFT_KEYS:
        g10["p_rows_" + k] = w[SAMPLE_ROWS, 0, :]; g10["m_rows_" + k] = m_[SAMPLE_ROWS, 0, :]; g10["v_rows_" + k] = v_[SAMPLE_ROWS, 0, :]
        g10["m_proj_" + k] = PROJ @ m_[:, 0, :].astype(np.float64); g10["v_proj_" + k] = PROJ @ v_[:, 0, :].astype(np.float64)
    else:
        g10["p_" + k] = w; g10["m_" + k] = m_; g10["v_" + k] = v_
# the oracle's state after the same three steps is what the reference saved
for i, k in enumerate(names):
    report("ckpt exp_avg " + k.replace("mpaec.", "")[:22], Mq[k], osd["state"][i]["exp_avg"].numpy(), 1e-3)      # the clip coefficient differs by ~1.5e-4 (summation order of the 4 M-element L1 norm)
    report("ckpt exp_avg_sq " + k.replace("mpaec.", "")[:19], Vq[k], osd["state"][i]["exp_avg_sq"].numpy(), 1e-3)

# the product's loader reads the REAL reference-written file (CPU side of the product: no kernels involved)
from signaltrain_amd import misc as my_misc, nn_proc as my_nn                                          # noqa: E402
my_nn._QUIET = True
sd, rv = my_misc.load_checkpoint(ck, device="cpu")
mine = my_nn.st_model(scale_factor=rv["scale_factor"], shrink_factor=rv["shrink_factor"], num_knobs=len(rv["knob_names"]), sr=rv["sr"])
mine.load_state_dict(sd)
for k, v in mine.state_dict().items():
    assert np.array_equal(v.numpy(), raw["state_dict"][k].numpy()), k
flat = my_misc.flatten_optimizer_state(rv["optimizer"], [tuple(v.shape) for v in sd.values()])
assert flat["step"] == 3 and abs(flat["lr"] - osd["param_groups"][0]["lr"]) < 1e-18
assert np.array_equal(flat["exp_avg"][1], osd["state"][1]["exp_avg"].numpy().ravel())
print("   signaltrain_amd.misc.load_checkpoint + st_model.load_state_dict + flatten_optimizer_state read the reference's file")
os.remove(ck); os.rmdir(tmp)

np.savez_compressed(os.path.join(OUT, "g4b_backward_clip.npz"), **g4b)
np.savez_compressed(os.path.join(OUT, "g5b_adam_clip.npz"), **g5b)
np.savez_compressed(os.path.join(OUT, "g10_checkpoint.npz"), **g10)
for f in ("g4b_backward_clip.npz", "g5b_adam_clip.npz", "g10_checkpoint.npz"):
    print(f"{f:28s} {os.path.getsize(os.path.join(OUT, f))/1024:8.1f} KiB")
print("golden capture (round 2) OK")
