#!/usr/bin/env python3
"""Capture golden vectors from the *imported reference* (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden.py
Writes tests/golden/*.npz (small, committed) and asserts that oracle/st_oracle.py reproduces
every captured value -- this is what pins the oracle.  The reference is imported
read-only from /root/reference with the in-process monkeypatches listed in SURVEY.md 8c; nothing
from it is copied into the repo.  This script never runs on the GPU box.
"""
import os, sys, types
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import scipy.signal, scipy.signal.windows
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/signaltrain"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
from oracle import st_oracle as O            # noqa: E402
from tests.golden_util import perturb_stft, sample_rows, projections, ae_keys, SAMPLE_ROWS  # noqa: E402

# ---- monkeypatches -------------------------------------------------------
scipy.signal.hamming = scipy.signal.windows.hamming
scipy.signal.cosine = scipy.signal.windows.cosine
torch.has_cudnn = False
nb = types.ModuleType("numba")
def _jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
nb.jit = _jit
sys.modules["numba"] = nb
sys.modules["librosa"] = types.ModuleType("librosa")
sys.path.insert(0, REF)
import nn_proc, loss_functions, learningrate, cls_fe_dct_bases      # noqa: E402  (the reference)
import audio as ref_audio                                            # noqa: E402

torch.set_num_threads(8)
os.makedirs(OUT, exist_ok=True)


def sd_numpy(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def report(name, a, b, tol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.max(np.abs(a - b)) if a.size else 0.0
    scale = max(np.max(np.abs(b)), 1e-30) if b.size else 1.0
    print(f"  {name:34s} max|d|={err:.3e} rel={err/scale:.3e}")
    assert err <= tol * max(scale, 1e-30) + 1e-30, f"oracle mismatch on {name}: {err} vs scale {scale}"


# ---- G1 geometry -------------------------------------------------------------------------
print("G1 geometry")
rows = []
for s in (1, 2, 4, 8):
    for sh in (1, 2, 4):
        for scheme in ("lean", "legacy"):
            if scheme == "legacy" and s == 8:
                continue                     # 268 M parameters; not instantiated here
            m = nn_proc.st_model(scale_factor=s, shrink_factor=sh, num_knobs=4, scale_scheme=scheme)
            T = m.mpaec.aenc._T; OT = m.mpaec.aenc._OT
            rows.append((s, sh, 0 if scheme == "lean" else 1, m.in_chunk_size, m.out_chunk_size, T, OT,
                         m.mpaec.dft_analysis.sz, m.mpaec.dft_analysis.hop))
            g = O.geometry(s, sh, scheme)
            assert (g["L"], g["y"], g["T"], g["OT"], g["N"], g["H"]) == rows[-1][3:], (g, rows[-1])
G1 = np.array(rows, np.int64)

# ---- G2 init bases -----------------------------------------------------------------------
print("G2 init bases")
torch.manual_seed(218); np.random.seed(218)
model = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=4)
geo = O.geometry(1, 4)
sd0 = sd_numpy(model)
P0 = O.init_params(geo, 4)
g2 = {}
for k in O.STFT_KEYS:
    ref_w = sd0[k][:, 0, :]
    report(k.split(".")[-2], P0[k][:, 0, :], ref_w, 2e-7)
    g2["rows_" + k] = ref_w[SAMPLE_ROWS]
    g2["colsum_" + k] = ref_w.astype(np.float64).sum(0)
g2["hamming"] = scipy.signal.windows.hamming(1024)
g2["gla"] = nn_proc.Synthesis.GLA(1024, 384, 1024)
report("hamming", O.hamming(1024), g2["hamming"], 1e-15)
report("gla", O.gla_window(1024, 384), g2["gla"], 1e-14)

# ---- inputs: comp_4c-shaped windows (oracle's generator; inputs are just inputs) -------------
rng = np.random.default_rng(218)
B = 2
X, Y, KN = O.synth_comp4c_batch(B, geo["L"], geo["y"], rng)

# ---- G9 compressor restatement vs reference audio.compressor_4controls ---------------------
print("G9 compressor")
xs = X[0].astype(np.float32)
yr = ref_audio.compressor_4controls(xs.copy(), -18.0, 3.0, 0.004, 0.02, 44100.0)
yo = O.compressor_4controls(xs.copy(), -18.0, 3.0, 0.004, 0.02, 44100.0)
report("compressor_4controls", yo, yr, 1e-6)

# ---- "learned" STFT weights: init + portable integer-hash perturbation ----------------------
ae_sd = {k: sd0[k] for k in ae_keys()}
P = dict(P0)
for k in ae_keys():
    P[k] = sd0[k]
perturb_stft(P, seed=7)
with torch.no_grad():
    model.load_state_dict({k: torch.from_numpy(P[k]) for k in P})

# ---- G3 forward --------------------------------------------------------------------------
print("G3 forward")
xt, kt, yt = torch.from_numpy(X), torch.from_numpy(KN), torch.from_numpy(Y)
model.train()
y_hat, mag, mag_hat, acts = model.forward(xt, kt, return_acts=True)
oy, omag, omag_hat, oc = O.model_fwd(X, KN, P, geo, return_all=True)
report("y_hat", oy, y_hat.detach().numpy(), 3e-6)
report("mag", omag, mag.detach().numpy(), 3e-6)
report("mag_hat", omag_hat, mag_hat.detach().numpy(), 3e-6)
a = [t.detach().numpy() for t in acts]
report("re", oc["re"], a[0], 3e-6); report("im", oc["im"], a[1], 3e-6)
report("Are", oc["Are"], a[-4], 5e-6); report("Aim", oc["Aim"], a[-3], 5e-6)
report("x_fwdsyn", oc["syn"], a[-2], 5e-6)
# AE activations (post-ELU), subsampled bins to keep the fixture small
FB = np.arange(0, 513, 19)
g3 = dict(x=X, knobs=KN, y=Y, y_hat=y_hat.detach().numpy(), mag=a[2], phs=a[3], re=a[0], im=a[1],
          mag_hat=mag_hat.detach().numpy(), phs_hat=a[-5], an_real=a[-4], an_imag=a[-3],
          x_fwdsyn=a[-2], act_bins=FB)
# acts layout: [re, im, mag, phs] + 10 (mag AE) + 10 (phs AE) + [mag_hat, phs_hat, an_real, an_imag, x_fwdsyn, y_hat]
for j in range(10):
    g3[f"m_act{j}"] = a[4 + j][:, FB, :]
    g3[f"p_act{j}"] = a[14 + j][:, FB, :]
# oracle hs: hs[0]=input rows, hs[1..9]=post-ELU layers; reference acts: z1..z4, catted, z5..z8, out
for (pref, hs) in (("m", oc["hs_m"]), ("p", oc["hs_p"])):
    ref_l = [g3[f"{pref}_act{j}"] for j in range(10)]
    # reference acts: z1..z4, catted, z5..z8, out(after skip) ; oracle hs: in, h1..h3, [h4;knobs], h5..h8, e9
    for j in range(9):
        o = hs[j + 1] if j < 4 else hs[j]
        if j == 3:
            o = o[:, :, :16]
        # phase rows inherit atan2's conditioning near |re|,|im| ~ 0 (SURVEY.md section 7 hard parts)
        report(f"{pref}-AE act{j}", o[:, FB, :], ref_l[j], 5e-6 if pref == "m" else 5e-5)

# ---- G4 backward ---------------------------------------------------------------------------
print("G4 backward")
F = geo["F"]
sbf = torch.exp((7. / F) * torch.arange(0., F)).expand_as(mag_hat).float()
loss = loss_functions.calc_loss(y_hat.float(), yt.float(), mag_hat.float(), scale_by_freq=sbf)
model.zero_grad()
loss.backward()
gref = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
# float64 oracle for gradient structure, float32 oracle for the fp32 path
X64, KN64, Y64 = X.astype(np.float64), KN.astype(np.float64), Y.astype(np.float64)
ol64, og64, _ = O.model_loss_bwd(X64, KN64, Y64, P, geo)
ol32, og32, oc32 = O.model_loss_bwd(X, KN, Y, P, geo)
report("loss (f64 oracle)", ol64, loss.item(), 3e-5)
report("loss (f32 oracle)", ol32, loss.item(), 3e-5)
for k in gref:
    report("grad " + k.replace("mpaec.", "")[:28], og64[k], gref[k], 2e-4)
g4 = dict(loss=np.float64(loss.item()))
for k in ae_keys():
    g4["g_" + k] = gref[k]
PROJ = projections(seed=11)
for k in O.STFT_KEYS:
    g = gref[k][:, 0, :]
    g4["l1_" + k] = np.float64(np.abs(g.astype(np.float64)).sum())
    g4["rows_" + k] = g[SAMPLE_ROWS]
    g4["cols_" + k] = g[:, SAMPLE_ROWS]
    g4["proj_" + k] = PROJ @ g.astype(np.float64)
norm_ref = float(sum(np.abs(gref[k].astype(np.float64)).sum() for k in O.STFT_KEYS))
total_norm = torch.nn.utils.clip_grad_norm_(
    list(model.mpaec.dft_analysis.parameters()) + list(model.mpaec.dft_synthesis.parameters()),
    max_norm=1., norm_type=1)
g4["clip_norm"] = np.float64(total_norm.item())
g4["clip_coef"] = np.float64(min(1.0, 1.0 / (total_norm.item() + 1e-6)))
og_c = {k: v.copy() for k, v in og32.items()}
n_o, c_o = O.clip_l1_stft(og_c)
report("clip norm", n_o, g4["clip_norm"], 1e-3)   # sum|g| over 4M fp32 values: order-sensitive at ~2e-4
print("   clip coef ref", g4["clip_coef"], "oracle", c_o)

# ---- G5 three Adam steps (train.py:131-151 ordering, lr applied after the step) ------------
print("G5 adam x3")
with torch.no_grad():
    model.load_state_dict({k: torch.from_numpy(P[k]) for k in P})
lrs, _ = learningrate.get_1cycle_schedule(lr_max=1e-3, n_data_points=200, epochs=1, batch_size=B)
opt = torch.optim.Adam(list(model.parameters()), lr=lrs[0], weight_decay=0)
Pq = {k: P[k].copy() for k in O.param_order()}
Mq = {k: np.zeros_like(v) for k, v in Pq.items()}
Vq = {k: np.zeros_like(v) for k, v in Pq.items()}
g5 = dict(lrs=lrs[:4].copy())
for it in range(3):
    Xi = np.roll(X, 17 * it, axis=1).copy(); Yi = np.roll(Y, 17 * it, axis=1).copy()
    lr_used = opt.param_groups[0]["lr"]
    yh, mg, mh = model.forward(torch.from_numpy(Xi), kt)
    ls = loss_functions.calc_loss(yh.float(), torch.from_numpy(Yi).float(), mh.float(), scale_by_freq=sbf)
    opt.zero_grad(); ls.backward(); model.clip_grad_norm_(); opt.step()
    opt.param_groups[0]["lr"] = lrs[it]                      # train.py:150
    lo, no, co = O.train_step(Xi, KN, Yi, Pq, Mq, Vq, it + 1, lr_used, geo)
    sdn = sd_numpy(model)
    g5[f"loss{it}"] = np.float64(ls.item()); g5[f"lr_used{it}"] = np.float64(lr_used)
    report(f"step{it} loss", lo, ls.item(), 3e-5)
    for k in ae_keys():
        g5[f"s{it}_" + k] = sdn[k]
    worst = 0.0
    for k in O.param_order():
        d = np.max(np.abs(Pq[k].astype(np.float64) - sdn[k]))
        worst = max(worst, d)
    print(f"  step{it} max |param diff| oracle vs ref = {worst:.3e}")
    assert worst < 5e-6
    for k in O.STFT_KEYS:
        g5[f"s{it}_rows_" + k] = sdn[k][SAMPLE_ROWS, 0, :]
        g5[f"s{it}_proj_" + k] = PROJ @ sdn[k][:, 0, :].astype(np.float64)

# ---- G6 1-cycle table ----------------------------------------------------------------------
print("G6 1cycle")
lr6, mom6 = learningrate.get_1cycle_schedule(lr_max=1e-4, n_data_points=200000, epochs=100, batch_size=200)
o6, om6 = O.get_1cycle_schedule(lr_max=1e-4, n_data_points=200000, epochs=100, batch_size=200)
assert len(lr6) == len(o6)
report("lr table", o6, lr6, 1e-15); report("mom table", om6, mom6, 1e-15)
idx6 = np.unique(np.concatenate([np.arange(0, len(lr6), 997), [len(lr6) - 1]]))
g6 = dict(idx=idx6, lr=lr6[idx6], mom=mom6[idx6], n=np.int64(len(lr6)))

# ---- G7 DCT-variant module I/O -------------------------------------------------------------
print("G7 dct bases")
dan = cls_fe_dct_bases.Analysis(); dsy = cls_fe_dct_bases.Synthesis()
Wd = dan.conv_analysis.weight.detach().numpy()[:, 0, :]
bd = dan.conv_analysis.bias.detach().numpy()
report("dct basis", O.dct_bases(), Wd, 3e-7)
xd = X[:1]
with torch.no_grad():
    xft = torch.transpose(dan.conv_analysis(torch.from_numpy(xd).view(1, 1, -1)), 2, 1)
    wav = dsy.forward(xft)
o_xft = O.dct_analysis_fwd(xd, Wd, bd)
report("dct analysis", o_xft, xft.numpy(), 5e-6)
o_wav = O.dct_synthesis_fwd(xft.numpy(), dsy.conv_synthesis.weight.detach().numpy()[:, 0, :])
report("dct synthesis", o_wav, wav.numpy(), 5e-6)
g7 = dict(x=xd, bias=bd, basis_rows=Wd[SAMPLE_ROWS], xft=xft.numpy()[:, :, ::8], wav=wav.numpy())

# ---- G8 scale=8 lean forward ---------------------------------------------------------------
print("G8 scale=8 lean forward")
torch.manual_seed(218)
m8 = nn_proc.st_model(scale_factor=8, shrink_factor=4, num_knobs=4)
geo8 = O.geometry(8, 4)
sd8 = sd_numpy(m8)
P8 = O.init_params(geo8, 4)
for k in ae_keys():
    P8[k] = sd8[k]
perturb_stft(P8, seed=9)
with torch.no_grad():
    m8.load_state_dict({k: torch.from_numpy(P8[k]) for k in P8})
rng8 = np.random.default_rng(8)
X8, Y8, KN8 = O.synth_comp4c_batch(1, geo8["L"], geo8["y"], rng8)
with torch.no_grad():
    y8, mg8, mh8 = m8.forward(torch.from_numpy(X8), torch.from_numpy(KN8))
oy8, omg8, omh8 = O.model_fwd(X8, KN8, P8, geo8)
report("s8 y_hat", oy8, y8.numpy(), 5e-6); report("s8 mag_hat", omh8, mh8.numpy(), 5e-6)
g8 = dict(x=X8, knobs=KN8, y_hat=y8.numpy(), mag_hat=mh8.numpy()[:, :, ::4])
for k in ae_keys():
    g8["ae_" + k] = sd8[k]

# ---- write ---------------------------------------------------------------------------------
np.savez_compressed(os.path.join(OUT, "g1_geometry.npz"), table=G1)
np.savez_compressed(os.path.join(OUT, "g2_init_bases.npz"), **g2)
np.savez_compressed(os.path.join(OUT, "g3_forward.npz"), **g3, **{"ae_" + k: ae_sd[k] for k in ae_sd})
np.savez_compressed(os.path.join(OUT, "g4_backward.npz"), **g4)
np.savez_compressed(os.path.join(OUT, "g5_adam.npz"), **g5)
np.savez_compressed(os.path.join(OUT, "g6_1cycle.npz"), **g6)
np.savez_compressed(os.path.join(OUT, "g7_dct.npz"), **g7)
np.savez_compressed(os.path.join(OUT, "g8_scale8.npz"), **g8)
np.savez_compressed(os.path.join(OUT, "g9_compressor.npz"), x=xs, y=yr,
                    knobs=np.array([-18.0, 3.0, 0.004, 0.02, 44100.0]))
for f in sorted(os.listdir(OUT)):
    print(f"{f:28s} {os.path.getsize(os.path.join(OUT, f))/1024:8.1f} KiB")
print("golden capture OK")
