"""CPU oracle for the SignalTrain training hot path -- TEST INFRASTRUCTURE ONLY.

This is a plain-numpy restatement of the reference algorithm (drscotthawley/signaltrain).
It is the *checker* for the HIP path in ``signaltrain_amd``; it is never imported by the
product package.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

Parity status: PINNED.  The reference ships no tests/golden vectors, so
the oracle is pinned against outputs of the reference itself: ``tools/capture_golden.py``
imports /root/reference (in the build container only), runs its forward / autograd
backward / clip / Adam on seeded inputs, writes ``tests/golden/*.npz`` and asserts this file
reproduces them (``tests/test_oracle_golden.py`` re-checks on every run, without the reference).

Every function cites the reference file:line it restates.  Shapes use the reference's
names: B windows, L samples/window, N=ft_size, H=hop, T input frames, OT output frames,
F=N/2+1 bins, K knobs, y output samples.
"""
import math
import numpy as np

EPS_ATAN = 1e-7          # nn_proc.py:310
L1_LAMBDA = 2e-5         # loss_functions.py:26
AE_LAYERS = ("fnn_enc", "fnn_enc2", "fnn_enc3", "fnn_enc4", "fnn_addknobs",
             "fnn_dec4", "fnn_dec3", "fnn_dec2", "fnn_dec")      # nn_proc.py:64-65


# ----------------------------------------------------------------------------- geometry
# ----------------------------------------------------------------------------- mixed-precision emulation
# GEMM_ROUND = None: exact arithmetic in the working dtype (the reference).  GEMM_ROUND = bf16_round: both operands of
# every analysis / synthesis GEMM (forward, data gradient, weight gradient) are rounded to bfloat16 first -- the
# arithmetic of the HIP library under st_set_precision(1) (bf16 operands, fp32 accumulation).
GEMM_ROUND = None


def bf16_round(a):
    """Round-to-nearest-even to bfloat16 (via float32, like the device), returned in the input dtype."""
    a = np.asarray(a)
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).astype(a.dtype).reshape(a.shape)


def fp16_round(a):
    """Round-to-nearest-even to IEEE float16 (out-of-range -> inf, as under the reference's Apex amp), returned in the input
    dtype: the operand conversion of the HIP library's ST_PREC_F16* levels (st_pack_f16 / pack_h4<2>)."""
    a = np.asarray(a)
    with np.errstate(over="ignore"):
        return a.astype(np.float16).astype(a.dtype)


def _r(a):
    return a if GEMM_ROUND is None else GEMM_ROUND(a)


# LOSS_SCALE = S: d loss is multiplied by S before the backward (Apex amp.scale_loss, train.py:134-135), so every gradient
# returned by model_loss_bwd carries S (what the rounding of the fp16 gradient operands sees); train_step divides it out
# before the clip, like the library's optimizer kernel.  Powers of two keep the scaling itself exact.
LOSS_SCALE = 1.0
# CLIP_ALL: the L1 clip runs over ALL parameters (train.py:136, the reference's Apex branch) instead of the 4 STFT tensors.
CLIP_ALL = False


# AE_ROUND = bf16_round: weights and layer inputs of the nine Linear layers of both autoencoders (forward, data gradient,
# weight gradient) are rounded to bfloat16 first -- the arithmetic of st_set_precision(2) (the BF instantiations of st_ae.h).
AE_ROUND = None


def _ra(a):
    return a if AE_ROUND is None else AE_ROUND(a)


def geometry(scale_factor=1, shrink_factor=4, scale_scheme="lean"):
    """st_model.__init__ geometry, nn_proc.py:357-385.  All integer, exact."""
    chunk = int(8192 * scale_factor)
    out_chunk = int(chunk / shrink_factor)
    ft, hop = 1024, 384
    if scale_scheme != "lean":                       # nn_proc.py:374-376
        ft, hop = int(ft * scale_factor), int(hop * scale_factor)
    T = int(np.ceil(chunk / float(hop)) + np.ceil(ft / float(hop)))
    OT = int(np.ceil(out_chunk / float(hop)) + np.ceil(ft / float(hop)))
    y = (OT - 1) * hop - ft
    return dict(L=chunk, out_chunk_intended=out_chunk, N=ft, H=hop, T=T, OT=OT, y=y, F=ft // 2 + 1)


def ae_layer_shapes(T, OT, K, R=64):
    """AsymAutoEncoder.__init__, nn_proc.py:46-61: (out, in) per layer in AE_LAYERS order."""
    return [(R, T), (R // 2, R), (R // 4, R // 2), (R // 4, R // 4), (R // 4, R // 4 + K),
            (R // 4, R // 4), (R // 2, R // 4), (R, R // 2), (OT, R)]


# ----------------------------------------------------------------------------- init bases
def hamming(N):
    """scipy.signal.hamming(N) (symmetric), used at cls_fe_dft.py:38,148."""
    n = np.arange(N, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / (N - 1))


def gla_window(wsz, hop):
    """Synthesis.GLA (LSEE-MSTFT synthesis window), cls_fe_dft.py:133-163."""
    synw = hamming(wsz)
    prod = synw ** 2
    env = np.zeros(wsz)
    red = wsz // hop
    for k in range(-red, red + 1):
        ind = hop * k + np.arange(1, wsz + 1)
        valid = (ind > 0) & (ind <= wsz)
        env[ind[valid] - 1] += prod[np.arange(wsz)[valid]]
    return synw / env


def dft_bases(N, window):
    """fft(eye(N), norm='ortho') * window split into (real, imag) float32 [N,1,N];
    Analysis.initialize cls_fe_dft.py:36-48, Synthesis.initialize cls_fe_dft.py:87-100."""
    f = np.fft.fft(np.eye(N), norm="ortho")
    return ((np.real(f) * window).astype(np.float32)[:, None, :],
            (np.imag(f) * window).astype(np.float32)[:, None, :])


def cosine_window(M):
    """scipy.signal.cosine(M): sin(pi*(n+.5)/M); cls_fe_dct_bases.py:10,67."""
    return np.sin(np.pi / M * (np.arange(M) + 0.5))


def dct_bases(freq_subbands=1024, window_size=2048):
    """core_modulation ('scott_method'), cls_fe_dct_bases.py:57-97 -> float32 [C, W]."""
    w = cosine_window(window_size)
    kvec = np.arange(freq_subbands) + 0.5
    nvec = np.arange(window_size) + 0.5 + freq_subbands / 2
    cos_an = w * np.cos(np.pi / freq_subbands * kvec[:, None] * nvec) * np.sqrt(2.0 / freq_subbands)
    return cos_an.astype(np.float32)


def xavier_normal(rng, out_f, in_f):
    """torch.nn.init.xavier_normal_ semantics (std = sqrt(2/(in+out))), nn_proc.py:71-75.
    The RNG stream differs from torch's, so values are *not* comparable with a torch-seeded
    model; parity tests always load explicit weights."""
    std = math.sqrt(2.0 / (in_f + out_f))
    return (rng.standard_normal((out_f, in_f)) * std).astype(np.float32)


def init_params(geo, K, rng=None):
    """State-dict-shaped parameter dict (the 40 keys of SURVEY.md section 5), float32."""
    rng = rng or np.random.default_rng(218)
    N, H = geo["N"], geo["H"]
    p = {}
    ar, ai = dft_bases(N, hamming(N))
    sr, si = dft_bases(N, gla_window(N, H))
    p["mpaec.dft_analysis.conv_analysis_real.weight"] = ar
    p["mpaec.dft_analysis.conv_analysis_imag.weight"] = ai
    p["mpaec.dft_synthesis.conv_synthesis_real.weight"] = sr
    p["mpaec.dft_synthesis.conv_synthesis_imag.weight"] = si
    for ae in ("aenc", "phs_aenc"):
        for name, (o, i) in zip(AE_LAYERS, ae_layer_shapes(geo["T"], geo["OT"], K)):
            p[f"mpaec.{ae}.{name}.weight"] = xavier_normal(rng, o, i)
            p[f"mpaec.{ae}.{name}.bias"] = np.zeros(o, np.float32)
    return p


STFT_KEYS = ("mpaec.dft_analysis.conv_analysis_real.weight",
             "mpaec.dft_analysis.conv_analysis_imag.weight",
             "mpaec.dft_synthesis.conv_synthesis_real.weight",
             "mpaec.dft_synthesis.conv_synthesis_imag.weight")


def param_order(K=None):
    """state_dict()/parameters() order of the reference model (module registration order)."""
    keys = list(STFT_KEYS)
    for ae in ("aenc", "phs_aenc"):
        for name in AE_LAYERS:
            keys += [f"mpaec.{ae}.{name}.weight", f"mpaec.{ae}.{name}.bias"]
    return keys


# ----------------------------------------------------------------------------- framing
def frame_starts(T, H, pad):
    """Start sample of frame t in the unpadded signal: H*t - pad  (Conv1d stride H, padding
    pad=N; cls_fe_dft.py:28-31).  Integer, exact -- the 'bit-exact indexing' contract."""
    return H * np.arange(T, dtype=np.int64) - pad


def frames(x, N, H, T, pad=None):
    """fr[b,t,n] = x[b, H*t + n - pad], zero outside [0,L).  x:[B,L] -> [B,T,N]."""
    pad = N if pad is None else pad
    B, L = x.shape
    xp = np.zeros((B, L + 2 * pad + N), x.dtype)
    xp[:, pad:pad + L] = x
    idx = (H * np.arange(T))[:, None] + np.arange(N)[None, :]      # into padded signal
    return xp[:, idx]


def overlap_add(frs, H):
    """full[b, H*t + n] += frs[b,t,n]  (ConvTranspose1d stride H; cls_fe_dft.py:112)."""
    B, OT, N = frs.shape
    full = np.zeros((B, (OT - 1) * H + N), frs.dtype)
    for t in range(OT):
        full[:, H * t:H * t + N] += frs[:, t, :]
    return full


# ----------------------------------------------------------------------------- forward pieces
def analysis_fwd(x_half, Wr, Wi, geo):
    """Analysis.forward, cls_fe_dft.py:50-58.  x_half is already x/2 (nn_proc.py:307).
    Wr/Wi: [N,1,N]; only rows < F are used.  Returns re, im [B,T,F]."""
    N, H, T, F = geo["N"], geo["H"], geo["T"], geo["F"]
    fr = frames(x_half, N, H, T)
    re = _r(fr) @ _r(Wr[:F, 0, :].T.astype(x_half.dtype))
    im = _r(fr) @ _r(Wi[:F, 0, :].T.astype(x_half.dtype))
    return re, im


def polar_fwd(re, im):
    """nn_proc.py:309-310: mag = ||(re,im)||_2 ; phs = atan2(im, re+1e-7) (fp32 in the ref)."""
    mag = np.sqrt(re * re + im * im)
    phs = np.arctan2(im, re + re.dtype.type(EPS_ATAN))
    return mag, phs


def elu(a):
    return np.where(a > 0, a, np.expm1(np.minimum(a, 0)))


def elu_grad_from_out(h):
    """ELU'(a) expressed through h = ELU(a): 1 if h > 0 else h + 1  (= exp(a) for a <= 0)."""
    return np.where(h > 0, np.ones_like(h), h + 1)


def ae_fwd(v, knobs, P, prefix, mode):
    """AsymAutoEncoder.forward, nn_proc.py:77-126.
    v: [B,T,F] (mag or phs); knobs [B,K]; mode 'sf' (nn_proc.py:115) or '' (nn_proc.py:117).
    Works on rows (b,f) with the T frames as features (transpose at nn_proc.py:79).
    Returns out [B,OT,F] and the list of post-activation layer outputs hs (h0 = input rows)."""
    dt = v.dtype
    B, T, F = v.shape
    x_in = np.transpose(v, (0, 2, 1))                       # [B,F,T]
    hs = [x_in]
    h = x_in
    for li, name in enumerate(AE_LAYERS):
        W = P[f"{prefix}.{name}.weight"].astype(dt)
        b = P[f"{prefix}.{name}.bias"].astype(dt)
        if name == "fnn_addknobs":                           # nn_proc.py:92-96
            kn = np.broadcast_to(knobs[:, None, :].astype(dt), (B, F, knobs.shape[1]))
            h = np.concatenate([h, kn], axis=2)
            hs[-1] = h                                       # input of this layer incl. knobs
        a = _ra(h) @ _ra(W).T + b
        h = elu(a)
        hs.append(h)
    OT = h.shape[2]
    if mode == "sf":
        out = h * x_in[:, :, T - OT:]
    else:
        out = h
    return np.transpose(out, (0, 2, 1)), hs


def fold_synthesis(Sr, Si, F):
    """Hermitian fold of the synthesis bases (SURVEY.md 8a': S'r[k]=Sr[k]+Sr[N-k], S'i[k]=Si[k]-Si[N-k]
    for 1<=k<=F-2; k=0 and k=F-1 unpaired).  Equivalent to the flip/cat at cls_fe_dft.py:109-110."""
    N = Sr.shape[0]
    Sr2, Si2 = Sr[:, 0, :], Si[:, 0, :]
    fr, fi = Sr2[:F].copy(), Si2[:F].copy()
    k = np.arange(1, F - 1)
    fr[k] += Sr2[N - k]
    fi[k] -= Si2[N - k]
    return fr, fi


def synthesis_fwd(Are, Aim, Sr, Si, geo, folded=True):
    """Synthesis.forward, cls_fe_dft.py:102-115. Are/Aim [B,OT,F] -> wave [B,y]."""
    N, H, F = geo["N"], geo["H"], geo["F"]
    dt = Are.dtype
    if folded:
        fr, fi = fold_synthesis(Sr.astype(dt), Si.astype(dt), F)
        frs = _r(Are) @ _r(fr) + _r(Aim) @ _r(fi)
    else:                                                    # literal flip/cat formulation
        re_full = np.concatenate([Are, Are[:, :, 1:-1][:, :, ::-1]], axis=2)
        im_full = np.concatenate([Aim, -Aim[:, :, 1:-1][:, :, ::-1]], axis=2)
        frs = re_full @ Sr[:, 0, :].astype(dt) + im_full @ Si[:, 0, :].astype(dt)
    full = overlap_add(frs, H)
    return full[:, N:full.shape[1] - N]


def freq_weights(F, dt=np.float32):
    """train.py:115-117: scale_by_freq = exp(7/F * arange(F)) (float32 in the reference)."""
    return np.exp((np.float32(7.0) / np.float32(F)) * np.arange(F, dtype=np.float32)).astype(np.float32).astype(dt)


def logcosh(d):
    """log(cosh(d)) evaluated stably; loss_functions.py:9-10 computes it literally."""
    # evaluated in float64 whatever the input precision: the loss *value* of a float32 run is a mean of ~1e-4-sized
    # terms, and the cancellation in this form (or in the reference's literal log(cosh(x)), whose cosh rounds to 1+eps)
    # costs ~1e-7 absolute per term in float32, i.e. up to 1e-2 of the mean.  The gradient (tanh) is unaffected.
    a = np.abs(np.asarray(d, dtype=np.float64))
    return a + np.log1p(np.exp(-2 * a)) - math.log(2.0)


def calc_loss(y_hat, y, mag_hat, scale_by_freq=None, l1_lambda=L1_LAMBDA):
    """loss_functions.calc_loss default branch (reg_logcosh=False), loss_functions.py:26-36."""
    lc = np.mean(logcosh(y - y_hat))
    if scale_by_freq is None:
        return lc + l1_lambda * np.mean(np.abs(mag_hat))
    return lc + l1_lambda / 10 * np.mean(np.abs(mag_hat * scale_by_freq))


# ----------------------------------------------------------------------------- whole model
def model_fwd(x, knobs, P, geo, return_all=False):
    """AsymMPAEC.forward, nn_proc.py:305-340.  Returns (y_hat2, mag, mag_hat[, cache])."""
    dt = x.dtype
    T, OT, y = geo["T"], geo["OT"], geo["y"]
    re, im = analysis_fwd(x / 2, P[STFT_KEYS[0]], P[STFT_KEYS[1]], geo)
    mag, phs = polar_fwd(re, im)
    mag_hat, hs_m = ae_fwd(mag, knobs, P, "mpaec.aenc", "sf")
    e9p, hs_p = ae_fwd(phs, knobs, P, "mpaec.phs_aenc", "")
    phs_hat = e9p + phs[:, T - OT:, :]
    Are = mag_hat * np.cos(phs_hat)
    Aim = mag_hat * np.sin(phs_hat)
    syn = synthesis_fwd(Are, Aim, P[STFT_KEYS[2]], P[STFT_KEYS[3]], geo)
    y_hat = syn + x[:, x.shape[1] - y:] / 2
    out = 2 * y_hat
    if not return_all:
        return out, mag, mag_hat
    cache = dict(re=re, im=im, mag=mag, phs=phs, hs_m=hs_m, hs_p=hs_p, mag_hat=mag_hat,
                 phs_hat=phs_hat, Are=Are, Aim=Aim, syn=syn)
    return out, mag, mag_hat, cache


def _ae_bwd(dout, v, knobs, P, prefix, mode, hs):
    """Hand-derived backward of ae_fwd (SURVEY.md 8a' 'MLP' lines).  dout [B,OT,F].
    Returns dv [B,T,F] and {param: grad}."""
    dt = v.dtype
    B, T, F = v.shape
    OT = dout.shape[1]
    d_out = np.transpose(dout, (0, 2, 1))                    # [B,F,OT]
    x_in = np.transpose(v, (0, 2, 1))
    dv = np.zeros((B, F, T), dt)
    e9 = hs[-1]
    if mode == "sf":
        dh = d_out * x_in[:, :, T - OT:]
        dv[:, :, T - OT:] += d_out * e9
    else:
        dh = d_out
    grads = {}
    for li in range(len(AE_LAYERS) - 1, -1, -1):
        name = AE_LAYERS[li]
        W = P[f"{prefix}.{name}.weight"].astype(dt)
        h_out, h_in = hs[li + 1], hs[li]
        da = dh * elu_grad_from_out(h_out[:, :, :dh.shape[2]])   # hs[4] also carries the knob columns
        da2 = da.reshape(-1, da.shape[2])
        grads[f"{prefix}.{name}.weight"] = _ra(da2).T @ _ra(h_in.reshape(-1, h_in.shape[2]))
        # wide geometries (T > 32 or OT > 16: st_ae_wide.h) run layers 1 and 9 as GEMMs in which the bias gradient rides along as a row of ones,
        # i.e. it is the sum of the ROUNDED dA under AE_ROUND; everywhere else it is summed from the fp32 accumulators
        wide_gemm_layer = (T > 32 or OT > 16) and li in (0, len(AE_LAYERS) - 1)
        grads[f"{prefix}.{name}.bias"] = (_ra(da2) if wide_gemm_layer else da2).sum(0)
        dh = _ra(da) @ _ra(W)
        if name == "fnn_addknobs":
            # the knob columns of the concatenated input (nn_proc.py:92-93: knobs repeated over the rows of a window, then torch.cat):
            # their gradient is the sum over the window's rows -- what autograd hands to a knobs tensor that requires grad
            grads["__d_knobs__"] = dh[:, :, W.shape[0]:].sum(1)      # [B, K]
            dh = dh[:, :, :W.shape[0]]
    dv += dh
    return np.transpose(dv, (0, 2, 1)), grads


def model_loss_bwd(x, knobs, y_true, P, geo, scale_by_freq=True):
    """One forward + loss + full backward (train.py:112-138).  Returns loss, grads, cache."""
    dt = x.dtype
    N, H, T, OT, F, ysz = geo["N"], geo["H"], geo["T"], geo["OT"], geo["F"], geo["y"]
    B = x.shape[0]
    out, mag, mag_hat, c = model_fwd(x, knobs, P, geo, return_all=True)
    w = freq_weights(F, dt) if scale_by_freq else None
    loss = calc_loss(out, y_true.astype(dt), mag_hat, w)
    lam = dt.type((L1_LAMBDA / 10 if scale_by_freq else L1_LAMBDA) * LOSS_SCALE)
    # d loss / d out (times the loss scale, if any)
    dy = -np.tanh(y_true.astype(dt) - out) / dt.type(B * ysz) * dt.type(LOSS_SCALE)
    dsyn = 2 * dy
    dfull = np.zeros((B, (OT - 1) * H + N), dt)
    dfull[:, N:N + ysz] = dsyn
    idx = (H * np.arange(OT))[:, None] + np.arange(N)[None, :]
    dfrs = dfull[:, idx]                                     # [B,OT,N]
    Sr, Si = P[STFT_KEYS[2]].astype(dt), P[STFT_KEYS[3]].astype(dt)
    fr_, fi_ = fold_synthesis(Sr, Si, F)
    dAre = _r(dfrs) @ _r(fr_.T)
    dAim = _r(dfrs) @ _r(fi_.T)
    dfr = _r(c["Are"].reshape(-1, F).T) @ _r(dfrs.reshape(-1, N))    # [F,N]
    dfi = _r(c["Aim"].reshape(-1, F).T) @ _r(dfrs.reshape(-1, N))
    gSr = np.zeros((N, N), dt); gSi = np.zeros((N, N), dt)
    gSr[:F] = dfr; gSi[:F] = dfi
    k = np.arange(1, F - 1)
    gSr[N - k] = dfr[k]
    gSi[N - k] = -dfi[k]
    cosp, sinp = np.cos(c["phs_hat"]), np.sin(c["phs_hat"])
    wf = (w if w is not None else np.ones(F, dt))
    dmag_hat = dAre * cosp + dAim * sinp + lam / dt.type(B * OT * F) * np.sign(mag_hat) * wf
    dphs_hat = mag_hat * (-dAre * sinp + dAim * cosp)
    dmag, g_m = _ae_bwd(dmag_hat, c["mag"], knobs, P, "mpaec.aenc", "sf", c["hs_m"])
    dphs, g_p = _ae_bwd(dphs_hat, c["phs"], knobs, P, "mpaec.phs_aenc", "", c["hs_p"])
    dphs[:, T - OT:, :] += dphs_hat
    d_knobs = g_m.pop("__d_knobs__") + g_p.pop("__d_knobs__")       # both autoencoders read the same knob settings (nn_proc.py:332-333)
    re, im = c["re"], c["im"]
    rp = re + dt.type(EPS_ATAN)
    den = rp * rp + im * im
    safe = np.where(c["mag"] > 0, c["mag"], 1)
    inv = np.where(c["mag"] > 0, 1 / safe, 0)                # norm subgradient 0 at the origin
    dre = dmag * re * inv - dphs * im / den
    dim = dmag * im * inv + dphs * rp / den
    if GEMM_ROUND is fp16_round:
        # fp16 configurations: the polar backward (fp32) saturates its result to the fp16 range before the weight-gradient GEMM
        # narrows it -- d atan2 ~ 1e7 on silent frames times the loss scale, and inf x 0 (the silent frame) would be NaN
        dre, dim = np.clip(dre, -65504.0, 65504.0), np.clip(dim, -65504.0, 65504.0)
    fr = frames(x / 2, N, H, T).reshape(-1, N)
    gWr = np.zeros((N, N), dt); gWi = np.zeros((N, N), dt)
    gWr[:F] = _r(dre.reshape(-1, F).T) @ _r(fr)
    gWi[:F] = _r(dim.reshape(-1, F).T) @ _r(fr)
    grads = {STFT_KEYS[0]: gWr[:, None, :], STFT_KEYS[1]: gWi[:, None, :],
             STFT_KEYS[2]: gSr[:, None, :], STFT_KEYS[3]: gSi[:, None, :]}
    grads.update(g_m); grads.update(g_p)
    c.update(dict(dy=dy, dAre=dAre, dAim=dAim, dmag_hat=dmag_hat, dphs_hat=dphs_hat,
                  dmag=dmag, dphs=dphs, dre=dre, dim=dim, out=out, d_knobs=d_knobs))
    return loss, grads, c


# ----------------------------------------------------------------------------- optimiser
def clip_l1_stft(grads, max_norm=1.0, all_params=None):
    """AsymMPAEC.clip_grad_norm_ (nn_proc.py:299-302): L1 norm over the 4 STFT tensors only,
    torch.nn.utils.clip_grad_norm_ semantics: coef = max_norm/(norm+1e-6), applied if < 1.
    all_params (default: CLIP_ALL): the Apex branch instead, train.py:136 -- the same clip over every parameter."""
    keys = list(grads.keys()) if (CLIP_ALL if all_params is None else all_params) else STFT_KEYS
    n = sum(np.abs(grads[k].astype(np.float64)).sum() for k in keys)
    n32 = np.float32(n)
    coef = np.float32(max_norm) / (n32 + np.float32(1e-6))
    if coef < 1:
        for k in keys:
            grads[k] = grads[k] * grads[k].dtype.type(coef)
    return float(n32), float(min(coef, np.float32(1.0)))


def adam_step(P, G, M, V, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor step (torch 2.x semantics; train.py:147,228):
    m=b1 m+(1-b1)g ; v=b2 v+(1-b2)g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    step_size = lr / bc1
    bc2s = math.sqrt(bc2)
    for k in P:
        g = G[k].astype(np.float32)
        M[k] = (M[k] + (g - M[k]) * np.float32(1 - b1)).astype(np.float32)   # lerp form used by torch
        V[k] = (V[k] * np.float32(b2) + np.float32(1 - b2) * g * g).astype(np.float32)
        denom = (np.sqrt(V[k]) / np.float32(bc2s) + np.float32(eps)).astype(np.float32)
        P[k] = (P[k] - np.float32(step_size) * (M[k] / denom)).astype(np.float32)


def get_1cycle_schedule(lr_max=1e-3, n_data_points=8000, epochs=200, batch_size=40):
    """learningrate.get_1cycle_schedule, learningrate.py:14-52."""
    pct_start, div_factor = 0.3, 15.0
    lr_start = lr_max / div_factor
    lr_end = lr_start / 1e2
    n_iter = n_data_points * epochs // batch_size
    a1 = int(n_iter * pct_start)
    a2 = n_iter - a1
    lrs_first = (lr_max - lr_start) * (1 - np.cos(np.linspace(0, np.pi, a1))) / 2 + lr_start
    lrs_second = (lr_max - lr_end) * (1 + np.cos(np.linspace(0, np.pi, a2))) / 2 + lr_end
    lrs = np.concatenate((lrs_first, lrs_second))
    mom_min, mom_max = 0.85, 0.95
    mom_avg, mom_amp = (mom_min + mom_max) / 2, (mom_max - mom_min) / 2
    moms = np.concatenate((mom_avg + mom_amp * np.cos(np.linspace(0, np.pi, a1)),
                           mom_avg - mom_amp * np.cos(np.linspace(0, np.pi, a2))))
    return lrs, moms


def train_step(x, knobs, y_true, P, M, V, step, lr, geo):
    """One optimisation step in the order of train.py:112-151 (forward, loss, backward,
    L1 clip of STFT grads, Adam).  `lr` is the value in param_groups at step time."""
    loss, grads, _ = model_loss_bwd(x, knobs, y_true, P, geo)
    if LOSS_SCALE != 1.0:
        grads = {k: g * g.dtype.type(1.0 / LOSS_SCALE) for k, g in grads.items()}      # amp unscales before the clip
    norm, coef = clip_l1_stft(grads)
    adam_step(P, grads, M, V, step, lr)
    return float(loss), norm, coef


# ----------------------------------------------------------------------------- DCT variant
def dct_analysis_fwd(x, W, bias, hop=1024, pad=1024):
    """cls_fe_dct_bases.Analysis.forward (:129-136): Conv1d(1->C, k=Wsz, stride=hop, padding=C)+bias,
    transposed to [B,T,C]."""
    C, Wsz = W.shape
    B, L = x.shape
    T = (L + 2 * pad - Wsz) // hop + 1
    fr = frames(x, Wsz, hop, T, pad=pad)
    return fr @ W.T.astype(x.dtype) + bias.astype(x.dtype)


def dct_synthesis_fwd(x_ft, W, hop=1024, crop=1024):
    """cls_fe_dct_bases.Synthesis.forward (:174-179): ConvTranspose1d(C->1, k=Wsz, stride=hop), crop C
    samples from each end.  x_ft [B,T,C] -> [B,1,len]."""
    frs = x_ft @ W.astype(x_ft.dtype)
    full = overlap_add(frs, hop)
    return full[:, None, crop:full.shape[1] - crop]


# ----------------------------------------------------------------------------- synthetic comp_4c data
def compressor_4controls(x, thresh=-24.0, ratio=2.0, attackTime=0.01, releaseTime=0.01, sr=44100.0):
    """audio.compressor_4controls, audio.py:380-426 (sequential attack/release smoother)."""
    N = len(x)
    dtype = x.dtype
    alphaA = np.exp(-np.log(9) / (sr * attackTime))
    alphaR = np.exp(-np.log(9) / (sr * releaseTime))
    x_dB = 20 * np.log10(np.abs(x) + 1e-8)
    x_dB = np.maximum(x_dB, -96)
    gc = np.zeros(N, dtype=dtype)
    i = x_dB > thresh
    gc[i] = thresh + (x_dB[i] - thresh) / ratio - x_dB[i]
    lin = np.zeros(N, dtype=dtype)
    prev = 0.0
    gcl = gc.tolist()
    out = [0.0] * N
    for n in range(1, N):
        g = gcl[n]
        if g < prev:
            prev = (1 - alphaA) * g + alphaA * prev
        else:
            prev = (1 - alphaR) * g + alphaR * prev
        out[n] = prev
    lin = np.power(10.0, np.asarray(out, dtype=dtype) / 20)
    return lin * x


COMP4C_RANGES = np.array([[-30, 0], [1, 5], [1e-3, 4e-2], [1e-3, 4e-2]])   # audio.py Compressor_4c.knob_ranges


def synth_comp4c_batch(B, L, ysz, rng, sr=44100.0, fast=True):
    """Synthetic comp_4c minibatch of the SynthAudioDataSet *shape and statistics*
    (datasets.py:312-334): x [B,L] f32, y [B,ysz] f32, knobs [B,4] f32 in [-.5,.5].
    Signals: random sines / noisy sines / decaying plucks / boxes (the chooser set {0,1,2,4,6,7}
    of datasets.py:317, simplified), effect = compressor_4controls.  `fast` uses a
    block-vectorised smoother (exact same recurrence evaluated with a python loop only when
    fast=False).  Used for benchmark / parity *inputs*; never for numerics claims about audio.py."""
    t = np.arange(L, dtype=np.float32) / sr
    X = np.zeros((B, L), np.float32); Y = np.zeros((B, ysz), np.float32)
    KN = (rng.beta(0.8, 0.8, size=(B, 4)) - 0.5).astype(np.float32)
    for b in range(B):
        ch = rng.choice([0, 1, 2, 4, 6, 7])
        if ch in (0, 1):
            s = np.zeros(L)
            for _ in range(rng.integers(1, 3)):
                s += rng.uniform(.2, .9) * np.cos(rng.uniform(5, 150) * (t - rng.random() * t[-1]))
            if ch == 1:
                s += 0.2 * rng.random() * (2 * rng.random(L) - 1)
        elif ch in (2, 7):
            s = np.zeros(L)
            for _ in range(rng.integers(1, 4)):
                s += rng.uniform(.5, .95) * rng.choice([-1, 1]) * np.sin(rng.uniform(50, 6400) * (t - rng.uniform(-.3, .3) * t[-1]))
            t0 = 0.35 * rng.random() * t[-1]
            env = np.exp(-12 * rng.random() * (t - t0)) * rng.uniform(.6, .95)
            env[t < t0] = rng.uniform(.1, .2)
            s = s * env
            if ch == 7:
                s += rng.uniform(.1, .4) * 0.1 * (2 * rng.random(L) - 1)
        else:
            s = np.full(L, rng.uniform(.1, .3))
            iu = int(0.3 * rng.random() * L); idn = min(iu + int(rng.uniform(.3, .65) * L), L - 1)
            s[:max(iu - 1, 0)] = 0.15 * rng.random(); s[iu:idn] = rng.uniform(.6, .95)
            if ch == 6:
                s = s * (2 * rng.random(L) - 1)
        m = np.max(np.abs(s)) + 1e-12
        if ch not in (4, 6):
            s = s / m * rng.uniform(.6, .9)
        s = (s * rng.choice([-1, 1]) + rng.random(L) * 1e-8).astype(np.float32)
        kv = COMP4C_RANGES[:, 0] + (KN[b] + 0.5) * (COMP4C_RANGES[:, 1] - COMP4C_RANGES[:, 0])
        yv = compressor_4controls(s, *kv, sr=sr) if not fast else _compressor_fast(s, *kv, sr=sr)
        if rng.random() < 0.5:
            s, yv = -s, -yv
        X[b] = s; Y[b] = yv[-ysz:]
    return X, Y, KN


def _compressor_fast(x, thresh, ratio, attackTime, releaseTime, sr=44100.0):
    """Same recurrence as compressor_4controls evaluated with scipy.signal.lfilter on
    attack/release runs would change rounding; instead run the exact scalar loop in float64
    over python lists (identical arithmetic, ~3 ms / window)."""
    return compressor_4controls(x, thresh, ratio, attackTime, releaseTime, sr)
