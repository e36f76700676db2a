"""GPU: the drop-in Python surface -- st_model through torch.autograd against the fused engine and the golden
forward fixture captured from the reference; Analysis/Synthesis modules; a short train.train() run."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _golden_model(golden_dir):
    from oracle import st_oracle as O
    from tests.golden_util import perturb_stft, ae_keys
    from signaltrain_amd import nn_proc
    nn_proc._QUIET = True
    g = np.load(os.path.join(golden_dir, "g3_forward.npz"))
    geo = O.geometry(1, 4)
    P = O.init_params(geo, 4)
    for k in ae_keys():
        P[k] = g["ae_" + k]
    perturb_stft(P, seed=7)
    m = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=4)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    return m.to("cuda:0"), g, P, geo


def test_golden_forward_through_st_model(golden_dir):
    """The reference's own forward outputs (golden G3) reproduced by the drop-in st_model on the GPU."""
    m, g, P, geo = _golden_model(golden_dir)
    x, kn = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["knobs"]).cuda()
    y, mag, mag_hat = m.forward(x, kn)
    for got, ref, name in ((y, g["y_hat"], "y_hat"), (mag, g["mag"], "mag"), (mag_hat, g["mag_hat"], "mag_hat")):
        e = np.abs(got.detach().cpu().numpy() - ref).max()
        assert e <= 1e-4 * np.abs(ref).max(), (name, e)          # north_star tolerance: 1e-4 relative, fp32
    # return_acts (nn_proc.py:311-338): all 30 tensors against the reference's own (golden G3; the 2 x 10 layer activations are stored at the
    # sampled bins `act_bins`).  They come from the HIP path's buffers / the library's diagnostic kernel, not from torch ops.
    y2, _, _, acts = m.forward(x, kn, return_acts=True)
    assert len(acts) == 30 and acts[0].shape == (2, 25, 513) and acts[-1].shape == (2, 2048)
    FB = g["act_bins"]
    refs = [g["re"], g["im"], g["mag"], g["phs"]] + [g[f"m_act{j}"] for j in range(10)] + [g[f"p_act{j}"] for j in range(10)] + \
           [g["mag_hat"], g["phs_hat"], g["an_real"], g["an_imag"], g["x_fwdsyn"], g["y_hat"] / 2]
    for i, (a, r) in enumerate(zip(acts, refs)):
        a = a.detach().cpu().numpy()
        if 4 <= i < 24:
            a = a[:, FB, :]
        assert a.shape == r.shape, (i, a.shape, r.shape)
        if i in (3, 25):                  # raw phases: atan2 is discontinuous at +-pi and ill-conditioned where mag ~ 0 -- compare on the unit circle, weighted by mag
            w = g["mag"] if i == 3 else g["mag_hat"]
            e = np.abs(w * (np.exp(1j * a) - np.exp(1j * r))).max()
            assert e <= 2e-4 * np.abs(w).max(), (i, e)
        else:
            assert np.abs(a - r).max() <= 2e-4 * max(np.abs(r).max(), 1e-12), (i, np.abs(a - r).max(), np.abs(r).max())


def test_autograd_matches_golden_backward(golden_dir):
    """loss.backward() through the custom autograd Function vs the reference's autograd (golden G4)."""
    from signaltrain_amd import loss_functions
    from tests.golden_util import ae_keys, projections, SAMPLE_ROWS, STFT_KEYS
    m, g, P, geo = _golden_model(golden_dir)
    g4 = np.load(os.path.join(golden_dir, "g4_backward.npz"))
    x, kn, yt = (torch.from_numpy(g[k]).cuda() for k in ("x", "knobs", "y"))
    y, mag, mag_hat = m.forward(x, kn)
    sbf = torch.exp((7. / 513) * torch.arange(0., 513, device="cuda")).expand_as(mag_hat).float()
    loss = loss_functions.calc_loss(y, yt, mag_hat, scale_by_freq=sbf)
    loss.backward()
    assert abs(loss.item() - float(g4["loss"])) <= 1e-4 * abs(float(g4["loss"]))
    grads = {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in m.named_parameters()}
    for k in ae_keys():
        ref = g4["g_" + k]
        assert np.abs(grads[k] - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-12, k
    PROJ = projections(seed=11)
    for k in STFT_KEYS:
        gk = grads[k][:, 0, :]
        sc = np.abs(g4["rows_" + k]).max()
        assert np.abs(gk[SAMPLE_ROWS] - g4["rows_" + k]).max() <= 2e-4 * sc
        assert np.abs(PROJ @ gk - g4["proj_" + k]).max() <= 2e-4 * np.abs(g4["proj_" + k]).max()
    # nn_proc.py:299-302 on the torch side: G4's norm is 0.216 (< 1), so the clip must leave the gradients untouched
    l1 = sum(float(dict(m.named_parameters())[k].grad.abs().sum()) for k in STFT_KEYS)
    assert abs(l1 - float(g4["clip_norm"])) <= 1e-3 * float(g4["clip_norm"])
    before = {k: dict(m.named_parameters())[k].grad.clone() for k in STFT_KEYS}
    m.clip_grad_norm_()
    for k in STFT_KEYS:
        assert torch.equal(dict(m.named_parameters())[k].grad, before[k]), k


def test_autograd_and_torch_clip_match_golden_active_clip(golden_dir):
    """Golden G4b (captured from the reference with clip_grad_norm_ ACTIVE: norm 2.71 -> coefficient 0.369): loss.backward()
    through the drop-in model, then model.clip_grad_norm_() (nn_proc.py:299-302) -- the clipped .grad of the four STFT tensors
    against the reference's clipped gradients, the autoencoder gradients untouched."""
    from signaltrain_amd import loss_functions
    from tests.golden_util import ae_keys, projections, SAMPLE_ROWS, STFT_KEYS
    m, g, P, geo = _golden_model(golden_dir)
    gb = np.load(os.path.join(golden_dir, "g4b_backward_clip.npz"))
    x, kn, yt = (torch.from_numpy(gb[k]).cuda() for k in ("x", "knobs", "y"))
    y, mag, mag_hat = m.forward(x, kn)
    sbf = torch.exp((7. / 513) * torch.arange(0., 513, device="cuda")).expand_as(mag_hat).float()
    loss = loss_functions.calc_loss(y, yt, mag_hat, scale_by_freq=sbf)
    loss.backward()
    assert abs(loss.item() - float(gb["loss"])) <= 1e-4 * abs(float(gb["loss"]))
    named = dict(m.named_parameters())
    l1 = sum(float(named[k].grad.abs().sum()) for k in STFT_KEYS)
    assert abs(l1 - float(gb["clip_norm"])) <= 1e-3 * float(gb["clip_norm"]) and l1 > 1.5
    m.clip_grad_norm_()
    PROJ = projections(seed=11)
    for k in STFT_KEYS:
        gk = named[k].grad.detach().cpu().numpy().astype(np.float64)[:, 0, :]
        sc = np.abs(gb["clipped_rows_" + k]).max()
        assert np.abs(gk[SAMPLE_ROWS] - gb["clipped_rows_" + k]).max() <= 1e-3 * sc, k
        assert np.abs(PROJ @ gk - gb["clipped_proj_" + k]).max() <= 1e-3 * np.abs(gb["clipped_proj_" + k]).max(), k
    l1c = sum(float(named[k].grad.abs().sum()) for k in STFT_KEYS)
    assert abs(l1c - 1.0) <= 2e-3                                    # clipped to max_norm = 1
    for k in ae_keys():
        ref = gb["g_" + k]
        assert np.abs(named[k].grad.detach().cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-12, k


def test_fused_step_matches_golden_active_clip(golden_dir):
    """Goldens G4b / G5b through the FUSED step (st_train_step: clip coefficient and Adam inside clip_adam_kernel): the norm and
    coefficient the kernel derived, and the parameters after each of three steps, against the reference's own run."""
    from signaltrain_amd.engine import StepEngine
    from tests.golden_util import ae_keys, projections, SAMPLE_ROWS, STFT_KEYS
    m, g, P, geo = _golden_model(golden_dir)
    gb = np.load(os.path.join(golden_dir, "g4b_backward_clip.npz")); g5 = np.load(os.path.join(golden_dir, "g5b_adam_clip.npz"))
    eng = StepEngine(m.engine(torch.zeros(3, 8192, device="cuda")).dims, "cuda:0"); eng.load_state_dict(P)
    kn = torch.from_numpy(gb["knobs"]).cuda()
    PROJ = projections(seed=11)
    for it in range(3):
        Xi = np.roll(gb["x"], 23 * it, axis=1).copy(); Yi = np.roll(gb["y"], 23 * it, axis=1).copy()
        eng.train_step(torch.from_numpy(Xi).cuda(), kn, torch.from_numpy(Yi).cuda(), float(g5[f"lr_used{it}"]))
        sc = eng.scalars.cpu().numpy()
        assert abs(sc[0] - float(g5[f"loss{it}"])) <= 1e-4 * abs(float(g5[f"loss{it}"])), it
        nref = float(g5[f"clip_norm{it}"])
        assert abs(sc[3] - nref) <= 1e-3 * nref and abs(sc[4] - 1.0 / (nref + 1e-6)) <= 1e-3 and sc[4] < 0.9, (it, sc[3], nref)
        if it == 0:
            assert abs(sc[3] - float(gb["clip_norm"])) <= 1e-3 * float(gb["clip_norm"]) and abs(sc[4] - float(gb["clip_coef"])) <= 1e-3
            # the clipped gradient is observable in the reference (p.grad after clip_grad_norm_): the fused step leaves it in grads
            for k in STFT_KEYS:
                gk = eng.named_grads[k].cpu().numpy().astype(np.float64)[:, 0, :]
                assert np.abs(gk[SAMPLE_ROWS] - gb["clipped_rows_" + k]).max() <= 1e-3 * np.abs(gb["clipped_rows_" + k]).max(), k
        for k in ae_keys():
            assert np.abs(eng.named[k].cpu().numpy() - g5[f"s{it}_" + k]).max() <= 5e-6, (it, k)
        for k in STFT_KEYS:
            w = eng.named[k].cpu().numpy()
            assert np.abs(w[SAMPLE_ROWS, 0, :] - g5[f"s{it}_rows_" + k]).max() <= 5e-6, (it, k)
            ref = g5[f"s{it}_proj_" + k]
            assert np.abs(PROJ @ w[:, 0, :].astype(np.float64) - ref).max() <= 1e-5 * np.abs(ref).max(), (it, k)


def test_second_forward_before_backward_is_safe(golden_dir):
    """the saved-for-backward state lives in the engine's single workspace; a validation forward (or another
    micro-batch) between forward and backward must not corrupt the gradients -- the stale stamp makes backward recompute."""
    from signaltrain_amd import loss_functions
    m, g, P, geo = _golden_model(golden_dir)
    x, kn, yt = (torch.from_numpy(g[k]).cuda() for k in ("x", "knobs", "y"))
    sbf = None
    def run(disturb):
        m.zero_grad()
        y, mag, mag_hat = m.forward(x, kn)
        if disturb:
            with torch.no_grad():
                m.forward(torch.flip(x, dims=[1]) * 0.3, -kn)        # another forward reuses the workspace
        sb = torch.exp((7. / 513) * torch.arange(0., 513, device="cuda")).expand_as(mag_hat).float()
        loss_functions.calc_loss(y, yt, mag_hat, scale_by_freq=sb).backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # gradient accumulation over two micro-batches == the sum of the separate gradients
    m.zero_grad()
    outs = [m.forward(x[i:i + 1], kn[i:i + 1]) for i in range(2)]
    tot = sum((o[0] * (i + 1.0)).sum() for i, o in enumerate(outs))
    tot.backward()
    acc = {k: p.grad.clone() for k, p in m.named_parameters()}
    ref = None
    for i in range(2):
        m.zero_grad()
        (m.forward(x[i:i + 1], kn[i:i + 1])[0] * (i + 1.0)).sum().backward()
        cur = {k: p.grad.clone() for k, p in m.named_parameters()}
        ref = cur if ref is None else {k: ref[k] + cur[k] for k in cur}
    for k in acc:
        assert (acc[k] - ref[k]).abs().max().item() <= 1e-6 * max(ref[k].abs().max().item(), 1e-12), k


def test_reference_style_loop_equals_fused_step(golden_dir):
    """zero_grad / backward / clip_grad_norm_ / torch.optim.Adam.step on the drop-in model == StepEngine.train_step."""
    from signaltrain_amd import loss_functions
    from signaltrain_amd.engine import StepEngine
    m, g, P, geo = _golden_model(golden_dir)
    x, kn, yt = (torch.from_numpy(g[k]).cuda() for k in ("x", "knobs", "y"))
    eng2 = StepEngine(m.engine(x).dims, "cuda:0"); eng2.load_state_dict(P)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0)
    for it in range(2):
        y, mag, mag_hat = m.forward(x, kn)
        sbf = torch.exp((7. / 513) * torch.arange(0., 513, device="cuda")).expand_as(mag_hat).float()
        loss = loss_functions.calc_loss(y, yt, mag_hat, scale_by_freq=sbf)
        opt.zero_grad(); loss.backward(); m.clip_grad_norm_(); opt.step()
        eng2.train_step(x, kn, yt, 1e-3)
    for k, v in m.state_dict().items():
        # Adam's m/(sqrt(v)+eps) turns ~1e-12 differences on noise-level gradient elements (|g| ~ eps) into a fraction of
        # lr; everything else agrees to fp32 rounding.  Bound: 10 % of one lr-sized update.
        assert (v - eng2.named[k]).abs().max().item() < 1e-4, k


def test_analysis_synthesis_modules_perfect_reconstruction():
    """cls_fe_dft.Analysis -> Synthesis at init is the identity (implicit invariant of the reference's init)."""
    from signaltrain_amd.cls_fe_dft import Analysis, Synthesis
    an, sy = Analysis().cuda(), Synthesis().cuda()
    x = (torch.randn(3, 8192, device="cuda") * 0.3)
    re, im = an(x)
    assert re.shape == (3, 25, 513)
    wave = sy(re, im)                      # all 25 frames -> (25-1)*384-1024 = 8192 samples
    assert wave.shape == (3, 8192)
    assert (wave - x).abs().max().item() < 1e-5
    (wave.square().mean()).backward()
    assert an.conv_analysis_real.weight.grad is not None and sy.conv_synthesis_imag.weight.grad.abs().max() > 0


def test_analysis_module_input_gradient():
    """The reference's Conv1d front end propagates d/d(wave) (cls_fe_dft.py:55-56 is plain autograd): Analysis.backward returns it too
    (needed only when something trainable sits upstream of the model) -- against float64 torch autograd of the same convolution."""
    import torch.nn.functional as Fn
    from signaltrain_amd.cls_fe_dft import Analysis
    an = Analysis().cuda()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    with torch.no_grad():
        for p in an.parameters(): p.add_(1e-3 * torch.randn(p.shape, device="cuda", generator=g))
    x = (torch.randn(2, 8192, device="cuda", generator=g) * 0.3).requires_grad_(True)
    re, im = an(x)
    cr, ci = torch.randn(re.shape, device="cuda", generator=g), torch.randn(im.shape, device="cuda", generator=g)
    ((re * cr).sum() + (im * ci).sum()).backward()
    xd = x.detach().double().cpu().requires_grad_(True)
    Wr, Wi = an.conv_analysis_real.weight.detach().double().cpu(), an.conv_analysis_imag.weight.detach().double().cpu()
    rr = Fn.conv1d(xd.unsqueeze(1), Wr, stride=384, padding=1024).transpose(1, 2)[:, :, :513]
    ii = Fn.conv1d(xd.unsqueeze(1), Wi, stride=384, padding=1024).transpose(1, 2)[:, :, :513]
    ((rr * cr.double().cpu()).sum() + (ii * ci.double().cpu()).sum()).backward()
    assert (re.detach().double().cpu() - rr.detach()).abs().max() < 1e-4 * rr.detach().abs().max()
    e = (x.grad.double().cpu() - xd.grad).abs().max() / xd.grad.abs().max()
    assert e < 1e-5, float(e)
    assert an.conv_analysis_real.weight.grad is not None


def test_train_driver_short_run(tmp_path):
    from signaltrain_amd import train, audio, nn_proc
    nn_proc._QUIET = True
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        model = train.train(effect=audio.Compressor_4c(), epochs=1, n_data_points=256, batch_size=32,
                            device=torch.device("cuda:0"), num_workers=2)
        assert os.path.isfile("modelcheckpoint.tar") and os.path.isfile("vl_avg_out.dat") and os.path.isfile("val_err_mae.dat")
        from signaltrain_amd import misc
        sd, rv = misc.load_checkpoint("modelcheckpoint.tar", device="cpu")
        assert len(sd) == 40 and rv["in_chunk_size"] == 8192 and rv["out_chunk_size"] == 2048
        assert all(torch.isfinite(v).all() for v in sd.values())
        # the 'optimizer' entry is a torch.optim.Adam state_dict (the reference's layout): torch loads it, and it is restored on resume
        ref_opt = torch.optim.Adam(model.parameters(), lr=1.0)
        ref_opt.load_state_dict(rv["optimizer"])
        steps0 = int(float(rv["optimizer"]["state"][0]["step"]))
        assert steps0 == 256 // 32 and rv["epoch"] == 1
        eng = model.engine(torch.zeros(32, 8192, device="cuda"))
        m_before = eng.m.clone()
        assert float(m_before.abs().max()) > 0
        model2 = train.train(effect=audio.Compressor_4c(), epochs=2, n_data_points=256, batch_size=32,
                             device=torch.device("cuda:0"), num_workers=2, resume_optimizer=True)        # resumes from modelcheckpoint.tar
        eng2 = model2.engine(torch.zeros(32, 8192, device="cuda"))
        assert eng2.step_count == 2 * steps0                          # epoch 2 only: optimizer step count and epoch counter continued
        sd2, rv2 = misc.load_checkpoint("modelcheckpoint.tar", device="cpu")
        assert rv2["epoch"] == 2 and int(float(rv2["optimizer"]["state"][0]["step"])) == 2 * steps0
        # the reference's behaviour (and the default): train `epochs` MORE from the loaded weights -- a finished run's checkpoint must not
        # turn the next train() call into a silent no-op
        w_before = {k: v.clone() for k, v in sd2.items()}
        model3 = train.train(effect=audio.Compressor_4c(), epochs=1, n_data_points=256, batch_size=32, device=torch.device("cuda:0"), num_workers=2)
        eng3 = model3.engine(torch.zeros(32, 8192, device="cuda"))
        assert eng3.step_count == steps0                              # fresh optimizer, one epoch of steps
        sd3, rv3 = misc.load_checkpoint("modelcheckpoint.tar", device="cpu")
        assert any(not torch.equal(sd3[k], w_before[k]) for k in sd3), "train() from a finished checkpoint trained nothing"
        # ... and with resume_optimizer=True a checkpoint whose position lies outside the new schedule keeps the moments but restarts the position
        model4 = train.train(effect=audio.Compressor_4c(), epochs=1, n_data_points=256, batch_size=32, device=torch.device("cuda:0"), num_workers=2,
                             resume_optimizer=True)
        assert model4.engine(torch.zeros(32, 8192, device="cuda")).step_count == 2 * steps0     # moments + their step counter kept, 8 more steps
    finally:
        os.chdir(cwd)


def test_scale8_golden_backward(golden_dir):
    """Golden G8b (the reference's autograd at the 65536-sample window, B = 1): loss and gradients of the HIP path -- the wide autoencoder
    kernels of st_ae_wide.h, the 174-frame GEMMs -- through the drop-in st_model + autograd AND through the fused loss_backward entry."""
    from oracle import st_oracle as O
    from tests.test_oracle_golden import golden_params
    from tests.golden_util import ae_keys, projections, SAMPLE_ROWS, STFT_KEYS
    from signaltrain_amd import nn_proc, loss_functions
    nn_proc._QUIET = True
    g8 = np.load(os.path.join(golden_dir, "g8_scale8.npz")); g = np.load(os.path.join(golden_dir, "g8b_scale8_backward.npz"))
    geo = O.geometry(8, 4)
    P = golden_params(golden_dir, geo, "g8_scale8.npz", "ae_", seed=9)
    m = nn_proc.st_model(scale_factor=8, shrink_factor=4, num_knobs=4)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})
    m = m.to("cuda:0")
    x, kn, yt = torch.from_numpy(g8["x"]).cuda(), torch.from_numpy(g8["knobs"]).cuda(), torch.from_numpy(g["y"]).cuda()
    PROJ = projections(seed=13)

    def check(loss, grads, what):
        assert abs(loss - float(g["loss"])) <= 1e-4 * abs(float(g["loss"])), (what, loss, float(g["loss"]))
        for k in ae_keys():
            ref = g["g_" + k]
            assert np.abs(grads[k] - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-12, (what, k)
        for k in STFT_KEYS:
            gk = grads[k].reshape(1024, 1024)
            assert np.abs(gk[SAMPLE_ROWS] - g["rows_" + k]).max() <= 2e-4 * np.abs(g["rows_" + k]).max(), (what, k)
            assert np.abs(PROJ @ gk - g["proj_" + k]).max() <= 2e-4 * np.abs(g["proj_" + k]).max(), (what, k)
            assert abs(np.abs(gk).sum() - float(g["l1_" + k])) <= 1e-3 * float(g["l1_" + k]), (what, k)
    y, mag, mag_hat = m.forward(x, kn)
    sbf = torch.exp((7. / 513) * torch.arange(0., 513, device="cuda")).expand_as(mag_hat).float()
    loss = loss_functions.calc_loss(y, yt, mag_hat, scale_by_freq=sbf)
    loss.backward()
    check(loss.item(), {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in m.named_parameters()}, "autograd")
    eng = m.engine(x)
    eng.loss_backward(x, kn, yt); torch.cuda.synchronize()
    check(float(eng.scalars[0]), {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.layout.views(eng.grads).items()}, "fused")


def test_scale8_golden_forward(golden_dir):
    """BASELINE configs[4] geometry (65536-sample window: T=174, OT=46, y=16256): the reference's forward
    outputs (golden G8, mag_hat sampled every 4th bin) reproduced by the HIP path through st_model."""
    from oracle import st_oracle as O
    from tests.test_oracle_golden import golden_params
    from signaltrain_amd import nn_proc
    nn_proc._QUIET = True
    g = np.load(os.path.join(golden_dir, "g8_scale8.npz"))
    geo = O.geometry(8, 4)
    P = golden_params(golden_dir, geo, "g8_scale8.npz", "ae_", seed=9)
    m = nn_proc.st_model(scale_factor=8, shrink_factor=4, num_knobs=g["knobs"].shape[1])
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})
    m = m.to("cuda:0")
    y, mag, mag_hat = m.forward(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["knobs"]).cuda())
    assert y.shape == (1, 16256) and mag.shape == (1, 174, 513) and mag_hat.shape == (1, 46, 513)
    for got, ref, name in ((y, g["y_hat"], "y_hat"), (mag_hat[:, :, ::4], g["mag_hat"], "mag_hat")):
        e = np.abs(got.detach().cpu().numpy() - ref).max()
        assert e <= 1e-4 * np.abs(ref).max(), (name, e)


@pytest.mark.parametrize("schedule,backend", [("two_bucket", "lib"), ("staged", "lib"), ("two_bucket", "torch"), ("staged", "torch")])
def test_dp_collective_path_single_rank(schedule, backend):
    """The N > 1 step (backward phases / stages, each followed by the RCCL all-reduce of the range it finalised, running under
    the next one -> st_dp_clip_adam) executed on ONE GPU with a single-rank RCCL communicator must reproduce the fused single-GPU
    step: through the library-owned communicator (st_dp_train_step, one C call) and through torch.distributed's collectives."""
    import socket
    import torch.distributed as dist
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    from signaltrain_amd.dp import DataParallel
    B, K = 4, 4
    geo, X, Y, KN, P = G.make_case(B, 21, K=K)
    d = G.dims_of(geo, B, K)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        e1 = StepEngine(d, G.DEV); e1.load_state_dict(P)
        e2 = StepEngine(d, G.DEV); e2.load_state_dict(P)
        dp = DataParallel(e2, force_collectives=True, schedule=schedule, backend=backend)
        assert dp.backend == backend
        dp.broadcast_parameters()
        x, kn, y = G.t(X), G.t(KN), G.t(Y)
        for it in range(3):
            e1.train_step(x, kn, y, 1e-3)
            dp.train_step(x, kn, y, 1e-3)
            torch.cuda.synchronize()
            l1, l2 = float(e1.scalars[0]), dp.mean_loss()
            assert abs(l1 - l2) <= 1e-5 * abs(l1), (it, l1, l2)
            err = (e1.params - e2.params).abs().max().item()
            assert err <= 2e-5, (it, err)          # same tolerance as the fused-vs-oracle parameter check (Adam amplifies ulp noise)
        dp.close()
    finally:
        dist.destroy_process_group()


def test_dp_packed_exchange_through_real_rccl_single_rank():
    """The bf16-packed last exchange (exchange flag 4, `DataParallel(pack16=True)`) through the REAL RCCL with one rank: ncclBfloat16 all-reduce of the packed analysis
    rows, widened by unstage_l1_kernel.  With one rank the only difference to the fused step is the bf16 rounding of those gradient rows: same loss, parameters
    within the 16-bit modes' step-level tolerance; in an fp32 engine the flag is ignored (bitwise the plain exchange)."""
    import socket
    import torch.distributed as dist
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    from signaltrain_amd.dp import DataParallel
    B, K = 4, 4
    geo, X, Y, KN, P = G.make_case(B, 21, K=K)
    d = G.dims_of(geo, B, K)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        x, kn, y = G.t(X), G.t(KN), G.t(Y)
        for dtype, tol in (("bf16_all", 3e-3), ("f32", 0.0)):
            e1 = StepEngine(d, G.DEV, compute_dtype=dtype); e1.load_state_dict(P)
            e2 = StepEngine(d, G.DEV, compute_dtype=dtype); e2.load_state_dict(P)
            ref = DataParallel(e1, force_collectives=True, schedule="staged", backend="lib")
            dp = DataParallel(e2, force_collectives=True, schedule="staged", backend="lib", pack16=True)
            for it in range(2):
                ref.train_step(x, kn, y, 1e-3); dp.train_step(x, kn, y, 1e-3)
                torch.cuda.synchronize()
                assert abs(ref.mean_loss() - dp.mean_loss()) <= 1e-3 * abs(ref.mean_loss()) + 1e-12
                err = (e1.params - e2.params).abs().max().item()
                assert err <= tol, (dtype, it, err)
            if dtype == "bf16_all":
                assert (e1.grads - e2.grads).abs().max().item() > 0          # the packed exchange really rounded the analysis rows
            ref.close(); dp.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scale,B", [(1, 5), (8, 2)])
def test_staged_backward_is_bitwise_the_fused_backward(scale, B):
    """the four data-parallel stages (st_loss_backward_stage) leave exactly the gradients and loss scalars
    of st_loss_backward -- the per-basis analysis GEMMs use another split-K count, but every slab sum is formed in the
    same order by the reduce kernel, so those are compared to fp32 reassociation tolerance and the rest bitwise; the four
    stage buckets are disjoint and cover every non-zero gradient."""
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    geo, X, Y, KN, P = G.make_case(B, 3, K=4, scale=scale)
    d = G.dims_of(geo, B, 4)
    e1 = StepEngine(d, G.DEV); e1.load_state_dict(P)
    e2 = StepEngine(d, G.DEV); e2.load_state_dict(P)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    e1.loss_backward(x, kn, y)
    for s in range(e2.N_STAGES):
        e2.loss_backward_stage(s, x, kn, y)
    torch.cuda.synchronize()
    o = e2.layout.offsets
    assert torch.equal(e1.grads[o[2]:], e2.grads[o[2]:])                        # synthesis bases + autoencoders: same kernels
    assert torch.equal(e1.scalars[:3], e2.scalars[:3])
    ga, gb = e1.grads[:o[2]], e2.grads[:o[2]]
    assert (ga - gb).abs().max().item() <= 1e-5 * ga.abs().max().item()         # analysis bases: split-K regrouping only
    cover = torch.zeros_like(e2.grads)
    for s in range(e2.N_STAGES):
        b = e2.stage_bucket(s)
        start = (b.data_ptr() - e2.grads.data_ptr()) // 4
        cover[start:start + b.numel()] += 1
    assert int(cover.max()) == 1
    assert int(((cover == 0) & (e2.grads != 0)).sum()) == 0


def test_device_compressor_matches_reference_golden(golden_dir):
    """st_compressor_4c (HIP) against the reference's own compressor_4controls output (golden G9),
    against the oracle on a batch of windows with different knob settings, and through SynthAudioDataSet.batch_device."""
    from oracle import st_oracle as O
    from signaltrain_amd import audio, datasets
    g = np.load(os.path.join(golden_dir, "g9_compressor.npz"))
    fx = audio.Compressor_4c(sr=float(g["knobs"][4]))
    x = torch.from_numpy(g["x"].astype(np.float32))[None].cuda()
    rng_ = fx.knob_ranges
    kn = torch.tensor([[(g["knobs"][i] - rng_[i, 0]) / (rng_[i, 1] - rng_[i, 0]) - 0.5 for i in range(4)]], dtype=torch.float32).cuda()
    y = fx.go_device(x, kn).cpu().numpy()[0]
    assert np.abs(y - g["y"]).max() <= 2e-6 * max(1.0, np.abs(g["y"]).max())
    # batch: several windows / knob settings / lengths incl. a ragged chunk boundary (L > 8192, not a multiple of the chunk)
    rng = np.random.default_rng(5)
    for L, ysz in ((8192, 2048), (20000, 20000), (65536, 16256)):
        B = 5
        X = (rng.standard_normal((B, L)) * np.linspace(0.02, 0.9, B)[:, None]).astype(np.float32)
        X[1, 100:900] = 0.0                                # silence: the -96 dB floor path
        KN = (rng.beta(0.8, 0.8, size=(B, 4)) - 0.5).astype(np.float32)
        fx = audio.Compressor_4c()
        yd = fx.go_device(torch.from_numpy(X).cuda(), torch.from_numpy(KN).cuda(), ysz).cpu().numpy()
        for b in range(B):
            kv = O.COMP4C_RANGES[:, 0] + (KN[b].astype(np.float64) + 0.5) * (O.COMP4C_RANGES[:, 1] - O.COMP4C_RANGES[:, 0])
            ref = audio.compressor_4controls(X[b], *kv, sr=44100.0)[-ysz:]          # gcc-built host helper (same float32 state rounding)
            assert np.abs(yd[b] - ref).max() <= 1e-5 * max(1e-3, np.abs(ref).max()), (L, b)
    ds = datasets.SynthAudioDataSet(8192, audio.Compressor_4c(), y_size=2048)
    xb, yb, kb = ds.batch_device(6)
    assert xb.shape == (6, 8192) and yb.shape == (6, 2048) and kb.shape == (6, 4) and torch.isfinite(yb).all()
    assert float(yb.abs().max()) <= float(xb.abs().max()) + 1e-6             # a downward compressor never amplifies


def test_predict_long_matches_windowed_oracle(golden_dir):
    """long-file inference (utils/predict_long.py) -- device-side framing + HIP forward against the
    oracle run window by window on the host (audio.sliding_window), including the zero-padded last window."""
    from oracle import st_oracle as O
    from signaltrain_amd import audio
    from signaltrain_amd.predict import predict_long
    m, g, P, geo = _golden_model(golden_dir)
    rng = np.random.default_rng(3)
    n = 8192 + 2048 * 6 + 777                                   # not a whole number of hops: exercises the tail padding
    sig = (0.4 * np.sin(np.arange(n) * 0.013) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    kn = np.array([0.1, -0.3, 0.25, -0.45], np.float32)
    y = predict_long(sig, kn, m, geo["L"], geo["y"], batch_size=3)
    # like the reference, the prediction starts after the first window's lookback: n - (chunk - out_chunk) samples
    assert y.shape == (n - (geo["L"] - geo["y"]),) and y.dtype == np.float32
    from oracle import host_audio
    xw = host_audio.sliding_window(sig, geo["L"], overlap=geo["L"] - geo["y"])
    ref = O.model_fwd(np.ascontiguousarray(xw), np.tile(kn, (xw.shape[0], 1)), P, geo)[0].reshape(-1)[:y.size]
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max()


def test_standalone_autoencoder_module(golden_dir):
    """AsymAutoEncoder.forward used on its own (nn_proc.py:77-126, modes 'sf' and '') against the oracle."""
    from oracle import st_oracle as O
    m, g, P, geo = _golden_model(golden_dir)
    rng = np.random.default_rng(8)
    v = np.abs(rng.standard_normal((2, geo["T"], geo["F"]))).astype(np.float32)
    kn = (rng.random((2, 4)) - 0.5).astype(np.float32)
    for mod, prefix, mode in ((m.mpaec.aenc, "mpaec.aenc", "sf"), (m.mpaec.phs_aenc, "mpaec.phs_aenc", "")):
        out, _ = mod.forward(torch.from_numpy(v).cuda(), torch.from_numpy(kn).cuda(), skip_connections=mode)
        ref = O.ae_fwd(v, kn, P, prefix, mode)[0]
        assert np.abs(out.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max(), mode


def test_dct_front_end_golden_and_autograd(golden_dir):
    """SURVEY.md row a15 (cls_fe_dct_bases.py): forward against the reference's golden G7; autograd (weight, bias and
    input gradients of both modules) against torch-CPU autograd of the same Conv1d / ConvTranspose1d with learned
    (perturbed) bases, odd batch."""
    import torch.nn.functional as Fn
    from signaltrain_amd import cls_fe_dct_bases as D
    g = np.load(os.path.join(golden_dir, "g7_dct.npz"))
    an, sy = D.Analysis().cuda(), D.Synthesis().cuda()
    with torch.no_grad():
        an.conv_analysis.bias.copy_(torch.from_numpy(g["bias"]))
    from tests.golden_util import SAMPLE_ROWS
    assert np.array_equal(an.conv_analysis.weight.detach().cpu().numpy()[SAMPLE_ROWS, 0], g["basis_rows"])
    xft = an.forward(g["x"])                                     # numpy input like the reference
    assert xft.shape == (1, 9, 1024)
    e = np.abs(xft.detach().cpu().numpy()[:, :, ::8] - g["xft"]).max()
    assert e <= 1e-4 * np.abs(g["xft"]).max(), e
    wav = sy.forward(xft)
    assert wav.shape == (1, 1, 8192)
    e = np.abs(wav.detach().cpu().numpy() - g["wav"]).max()
    assert e <= 1e-4 * np.abs(g["wav"]).max(), e
    # autograd vs torch CPU on perturbed bases
    rng = np.random.default_rng(4)
    B, L = 3, 8192
    Wa = (D.core_modulation(1024, 2048) + 0.01 * rng.standard_normal((1024, 2048))).astype(np.float32)
    Ws = (D.core_modulation(1024, 2048) + 0.01 * rng.standard_normal((1024, 2048))).astype(np.float32)
    bias = (0.1 * rng.standard_normal(1024)).astype(np.float32)
    x = (0.3 * rng.standard_normal((B, L))).astype(np.float32)
    proj = rng.standard_normal((B, 1, 8192)).astype(np.float32)
    # reference ops on CPU (float64)
    xc = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wac = torch.tensor(Wa[:, None, :], dtype=torch.float64, requires_grad=True); bc = torch.tensor(bias, dtype=torch.float64, requires_grad=True)
    Wsc = torch.tensor(Ws[:, None, :], dtype=torch.float64, requires_grad=True)
    ft = Fn.conv1d(xc[:, None, :], Wac, bc, stride=1024, padding=1024).transpose(2, 1)
    wv = Fn.conv_transpose1d(ft.transpose(2, 1), Wsc, stride=1024)[:, :, 1024:-1024]
    (wv * torch.tensor(proj, dtype=torch.float64)).sum().backward()
    # HIP path
    with torch.no_grad():
        an.conv_analysis.weight.copy_(torch.from_numpy(Wa[:, None, :])); an.conv_analysis.bias.copy_(torch.from_numpy(bias))
        sy.conv_synthesis.weight.copy_(torch.from_numpy(Ws[:, None, :]))
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    ftg = an.forward(xg); wvg = sy.forward(ftg)
    (wvg * torch.from_numpy(proj).cuda()).sum().backward()
    def chk(name, got, ref):
        ref = ref.detach().numpy(); e = np.abs(got.detach().cpu().numpy() - ref).max()
        assert e <= 1e-4 * np.abs(ref).max(), (name, e, np.abs(ref).max())
    chk("ft", ftg, ft); chk("wave", wvg, wv)
    chk("g_x", xg.grad, xc.grad); chk("g_Wa", an.conv_analysis.weight.grad, Wac.grad); chk("g_bias", an.conv_analysis.bias.grad, bc.grad)
    chk("g_Ws", sy.conv_synthesis.weight.grad, Wsc.grad)


def test_training_trajectory_matches_cpu_port():
    """25 consecutive optimisation steps (1-cycle learning rates, four alternating minibatches) on the GPU vs the
    PyTorch-CPU restatement of the reference's op sequence on identical data: the loss trajectory and the parameters
    after training.  (The 3-step oracle comparison pins the arithmetic; this pins the absence of drift.)"""
    from tests import gpu_checks as G
    from oracle.torch_cpu_step import CpuPort
    from signaltrain_amd.engine import StepEngine
    from signaltrain_amd import learningrate
    B, K, steps = 4, 4, 25
    geo, X, Y, KN, P = G.make_case(4 * B, 41, K=K)
    d = G.dims_of(geo, B, K)
    eng = StepEngine(d, G.DEV); eng.load_state_dict(P)
    port = CpuPort(P)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    lrs, _ = learningrate.get_1cycle_schedule(lr_max=2e-4, n_data_points=steps * B, epochs=1, batch_size=B)
    lg, lc = [], []
    for it in range(steps):
        sl = slice((it % 4) * B, (it % 4 + 1) * B)
        lr = float(lrs[max(it - 1, 0)])                                   # train.py:150: lr written for the NEXT step
        eng.train_step(G.t(X[sl]), G.t(KN[sl]), G.t(Y[sl]), lr)
        lg.append(float(eng.scalars[0]))
        lc.append(port.step(torch.from_numpy(X[sl]), torch.from_numpy(KN[sl]), torch.from_numpy(Y[sl]), lr))
    lg, lc = np.array(lg), np.array(lc)
    assert np.all(np.isfinite(lg))
    assert np.abs(lg - lc).max() <= 2e-3 * np.abs(lc).max(), (lg, lc)      # float32 log-cosh noise of the CPU port is ~1e-3
    worst = 0.0
    for k, v in eng.named.items():
        ref = port.P[k].detach().numpy().reshape(v.shape)
        worst = max(worst, float(np.abs(v.cpu().numpy() - ref).max()))
    assert worst <= 2e-4, worst                                            # 25 Adam steps of <= 2e-4 each: no drift beyond noise-level sign flips


def test_autograd_all_three_outputs(golden_dir):
    """loss = <y_hat, p1> + <mag, p2> + <mag_hat, p3> through st_model's autograd Function (exercises the upstream
    gradients of ALL outputs: g_y_hat, g_mag -> st_polar_bwd, g_mag_hat -> st_ae_bwd) vs float64 torch-CPU autograd of
    the reference op sequence (oracle/torch_cpu_step.forward)."""
    from oracle import torch_cpu_step as TC
    m, g, P, geo = _golden_model(golden_dir)
    rng = np.random.default_rng(12)
    B = 3
    x = (0.3 * rng.standard_normal((B, geo["L"]))).astype(np.float32)
    kn = (rng.random((B, 4)) - 0.5).astype(np.float32)
    p1 = rng.standard_normal((B, geo["y"])).astype(np.float32)
    p2 = (0.1 * rng.standard_normal((B, geo["T"], geo["F"]))).astype(np.float32)
    p3 = (0.1 * rng.standard_normal((B, geo["OT"], geo["F"]))).astype(np.float32)
    P64 = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    y, mg, mh = TC.forward(P64, torch.tensor(x, dtype=torch.float64), torch.tensor(kn, dtype=torch.float64))
    ((y * torch.tensor(p1, dtype=torch.float64)).sum() + (mg * torch.tensor(p2, dtype=torch.float64)).sum()
     + (mh * torch.tensor(p3, dtype=torch.float64)).sum()).backward()
    m.zero_grad()
    yg, mgg, mhg = m.forward(torch.from_numpy(x).cuda(), torch.from_numpy(kn).cuda())
    ((yg * torch.from_numpy(p1).cuda()).sum() + (mgg * torch.from_numpy(p2).cuda()).sum() + (mhg * torch.from_numpy(p3).cuda()).sum()).backward()
    sd = dict(m.named_parameters())
    worst = ("", 0.0)
    for k, ref in P64.items():
        got = sd[k].grad.detach().cpu().numpy().reshape(ref.shape)
        r = ref.grad.numpy()
        scale = np.abs(r).max()
        if "dft_" in k:                                            # rows >= F of the analysis tensors: exact zeros on both sides
            scale = max(scale, 1e-30)
        e = np.abs(got - r).max() / max(scale, 1e-30)
        if e > worst[1]:
            worst = (k, e)
        assert e <= 2e-4, (k, e)


def test_knob_gradient_matches_reference_autograd(golden_dir):
    """knobs.requires_grad_(): d loss / d knobs through the drop-in model (st_model_knob_grad: the exact per-window route) against the REFERENCE's own
    autograd (golden G12: the G3 / G4 inputs and weights, calc_loss with train.py's frequency weighting) -- and the parameter gradients of the same
    backward() still those of golden G4 (the knob passes must not leak into them)."""
    from signaltrain_amd import loss_functions
    from tests.golden_util import ae_keys
    m, g, P, geo = _golden_model(golden_dir)
    g4 = np.load(os.path.join(golden_dir, "g4_backward.npz")); g12 = np.load(os.path.join(golden_dir, "g12_knob_grad.npz"))
    x, yt = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    kn = torch.from_numpy(g["knobs"]).cuda().requires_grad_(True)
    y, mag, mag_hat = m.forward(x, kn)
    sbf = torch.exp((7. / 513) * torch.arange(0., 513, device="cuda")).expand_as(mag_hat).float()
    loss = loss_functions.calc_loss(y, yt, mag_hat, scale_by_freq=sbf)
    loss.backward()
    ref = g12["d_knobs"]
    got = kn.grad.detach().cpu().numpy().astype(np.float64)
    assert got.shape == ref.shape == (2, 4)
    assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max(), (got, ref)       # fp32 tolerance of the other gradients (G4)
    grads = {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in m.named_parameters()}
    for k in ae_keys():
        assert np.abs(grads[k] - g4["g_" + k]).max() <= 2e-4 * np.abs(g4["g_" + k]).max() + 1e-12, k


@pytest.mark.parametrize("dtype", ["f32", "bf16_all"])
def test_knob_gradient_all_three_outputs(golden_dir, dtype):
    """The knob gradient for upstream gradients of ALL outputs (<y_hat, p1> + <mag, p2> + <mag_hat, p3>; mag does not depend on the knobs), B = 3 with
    distinct windows, vs float64 torch-CPU autograd of the reference op sequence (oracle/torch_cpu_step.forward) -- and the per-window route is
    really per window: window 1's gradient is unchanged when the other windows of the batch change."""
    from oracle import torch_cpu_step as TC
    m, g, P, geo = _golden_model(golden_dir)
    m.set_compute_dtype(dtype)
    rng = np.random.default_rng(21)
    B = 3
    x = (0.3 * rng.standard_normal((B, geo["L"]))).astype(np.float32)
    kn = (rng.random((B, 4)) - 0.5).astype(np.float32)
    p1 = rng.standard_normal((B, geo["y"])).astype(np.float32)
    p2 = (0.1 * rng.standard_normal((B, geo["T"], geo["F"]))).astype(np.float32)
    p3 = (0.1 * rng.standard_normal((B, geo["OT"], geo["F"]))).astype(np.float32)
    P64 = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in P.items()}
    k64 = torch.tensor(kn, dtype=torch.float64, requires_grad=True)
    y, mg, mh = TC.forward(P64, torch.tensor(x, dtype=torch.float64), k64)
    ((y * torch.tensor(p1, dtype=torch.float64)).sum() + (mg * torch.tensor(p2, dtype=torch.float64)).sum()
     + (mh * torch.tensor(p3, dtype=torch.float64)).sum()).backward()
    ref = k64.grad.numpy()

    def run(xx):
        kg = torch.from_numpy(kn).cuda().requires_grad_(True)
        yg, mgg, mhg = m.forward(torch.from_numpy(xx).cuda(), kg)
        ((yg * torch.from_numpy(p1).cuda()).sum() + (mgg * torch.from_numpy(p2).cuda()).sum() + (mhg * torch.from_numpy(p3).cuda()).sum()).backward()
        return kg.grad.detach().cpu().numpy().astype(np.float64)
    got = run(x)
    tol = 2e-4 if dtype == "f32" else 6e-2           # 16-bit Linear layers: the tolerance class of the other bf16_all gradients (tests/gpu_checks.py)
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), (got, ref)
    x2 = x.copy(); x2[0] *= -0.5; x2[2] = x2[2][::-1]
    got2 = run(x2)
    assert np.array_equal(got2[1], got[1]) and not np.array_equal(got2[0], got[0])


def test_train_driver_device_feed(tmp_path):
    """train.train(device_feed=True): recycled synthetic dataset resident in HBM (effect computed on the GPU), random index
    gathers per minibatch -- no CPU workers in the loop; the loss must go down over a few epochs of a tiny dataset."""
    from signaltrain_amd import train, audio, nn_proc, datasets
    nn_proc._QUIET = True
    ds = datasets.DeviceRecycledDataSet(8192, audio.Compressor_4c(), datapoints=96, y_size=2048)
    assert ds.x.shape == (96, 8192) and ds.y.shape == (96, 2048) and ds.x.is_cuda
    seen = 0
    for x, y, k in ds.batches(32):
        assert x.shape == (32, 8192) and y.shape == (32, 2048) and k.shape == (32, 4)
        seen += 1
    assert seen == 3
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        torch.manual_seed(0); np.random.seed(0)
        model = train.train(effect=audio.Compressor_4c(), epochs=3, n_data_points=512, batch_size=32,
                            device=torch.device("cuda:0"), num_workers=2, device_feed=True, lr_max=2e-4)
        lines = [l.split() for l in open("vl_avg_out.dat").read().strip().splitlines()]
        assert len(lines) >= 3 and all(np.isfinite(float(l[-1])) for l in lines)
    finally:
        os.chdir(cwd)


def test_train_driver_host_feed_and_dataset_items(tmp_path):
    """The reference's own feed: train.train(device_feed=False) = a torch DataLoader with CPU workers over SynthAudioDataSet.__getitem__ (workers generate
    with the CPU-device form of the batched generators + the effect's host helper; the main process may use the GPU generators), and the Dataset contract on a GPU box:
    items out of a device-generated chunk, recycle=True fixed items."""
    from signaltrain_amd import train, audio, nn_proc, datasets
    nn_proc._QUIET = True
    np.random.seed(5)
    ds = datasets.SynthAudioDataSet(8192, audio.Compressor_4c(), datapoints=12, y_size=2048, item_chunk=8)
    x, y, k = ds[0]
    assert x.shape == (8192,) and y.shape == (2048,) and k.shape == (4,) and x.dtype == np.float32 and np.abs(x).max() > 0.05
    y2 = audio.Compressor_4c().go(x, k)[0][-2048:]                       # the target IS the effect of the input at those knobs (device compressor vs host helper)
    assert np.abs(y2 - y).max() < 2e-5 * max(1.0, np.abs(y2).max())
    rec = datasets.SynthAudioDataSet(8192, audio.Compressor_4c(), datapoints=5, y_size=2048, recycle=True, item_chunk=4)
    assert rec.x.shape == (5, 8192) and np.array_equal(rec[2][0], rec[2][0]) and not np.array_equal(rec[2][0], rec[3][0])
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        torch.manual_seed(0); np.random.seed(0)
        train.train(effect=audio.Compressor_4c(), epochs=1, n_data_points=64, batch_size=16, device=torch.device("cuda:0"), num_workers=2, device_feed=False, lr_max=2e-4)
        lines = [l.split() for l in open("vl_avg_out.dat").read().strip().splitlines()]
        assert len(lines) >= 1 and all(np.isfinite(float(l[-1])) for l in lines)
    finally:
        os.chdir(cwd)


def test_train_driver_bf16_all(tmp_path):
    """train.train(compute_dtype="bf16_all"): the mixed-precision step (bf16 operands in the STFT GEMMs and the autoencoder
    layers, fp32 master weights / optimizer) through the driver, device-resident data: its validation-loss trajectory over four
    epochs stays within 5 % of the fp32 run from the same seeds."""
    from signaltrain_amd import train, audio, nn_proc
    nn_proc._QUIET = True
    cwd = os.getcwd()
    traj = {}
    try:
        for dt in ("f32", "bf16_all"):
            sub = tmp_path / dt; sub.mkdir(); os.chdir(sub)
            torch.manual_seed(0); np.random.seed(0)
            train.train(effect=audio.Compressor_4c(), epochs=4, n_data_points=512, batch_size=32,
                        device=torch.device("cuda:0"), num_workers=2, device_feed="recycle", lr_max=2e-4, compute_dtype=dt)
            traj[dt] = np.array([float(l.split()[-1]) for l in open("vl_avg_out.dat").read().strip().splitlines()])
    finally:
        os.chdir(cwd)
    assert len(traj["f32"]) >= 4 and np.all(np.isfinite(traj["bf16_all"]))
    # epoch 1's entry is an un-debiased EMA over four validation batches taken after 16 steps (5.6 % apart on the windows of the fused
    # feed kernel); from epoch 2 on the two trajectories agree to ~1 %
    assert np.all(np.abs(traj["bf16_all"] - traj["f32"]) <= np.array([0.10] + [0.05] * (len(traj["f32"]) - 1)) * np.abs(traj["f32"])), traj


def test_clip_scope_explains_f16_loss():
    """VERDICT r2 weak #2 as an assertion (full table: profiles/r03_train_convergence.txt): the f16 modes end a short 1-cycle run ~40 % below
    fp32's validation loss.  That is the clip SCOPE of the reference's Apex branch (train.py:136: L1 clip over all parameters), not the fp16
    rounding: exact fp32 WITH that scope lands beside f16_all, and f16_all WITHOUT it lands beside bf16_all."""
    from signaltrain_amd import _lib, nn_proc, audio, datasets, learningrate
    from signaltrain_amd.engine import StepEngine
    nn_proc._QUIET = True
    dev = torch.device("cuda:0"); B, STEPS = 64, 600
    torch.manual_seed(218); np.random.seed(218)
    sd = {k: v.detach().clone() for k, v in nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=4).state_dict().items()}
    ds = datasets.SynthAudioDataSet(8192, audio.Compressor_4c(), datapoints=STEPS * B, y_size=2048)
    x, y, kn = ds.batch_device(STEPS * B, dev)
    xv, yv, kv = ds.batch_device(256, dev)
    lrs, _ = learningrate.get_1cycle_schedule(lr_max=1e-3, n_data_points=STEPS * B, epochs=1, batch_size=B)
    d, dv = _lib.geometry(1, 4, 4, B), _lib.geometry(1, 4, 4, 256)
    val = {}
    for mode, dt, ca in (("f32", "f32", None), ("f32/ca1", "f32", True), ("f16_all", "f16_all", None), ("f16_all/ca0", "f16_all", False), ("bf16_all", "bf16_all", None)):
        eng = StepEngine(d, dev, compute_dtype=dt, clip_all=ca); eng.load_state_dict(sd)
        for it in range(STEPS):
            sl = slice(it * B, (it + 1) * B)
            eng.train_step(x[sl], kn[sl], y[sl], float(lrs[max(it - 1, 0)]))
        assert int(eng.scalars[5]) == 0                                   # no overflow-skipped step at loss scale 4096
        ev = StepEngine(dv, dev); ev.params.copy_(eng.params)             # always evaluated in fp32
        ev.loss_backward(xv, kv, yv); torch.cuda.synchronize()
        val[mode] = float(ev.scalars[0])
    rel = {k: v / val["f32"] - 1.0 for k, v in val.items()}
    assert rel["f16_all"] < -0.2, rel                                     # the effect is there ...
    assert abs(rel["f32/ca1"] - rel["f16_all"]) < 0.12, rel               # ... exact fp32 with the Apex clip scope reproduces it ...
    # ... and fp16 without that scope does not show it (600 steps: -25 % / -12 % / -10 %; the 1000-step table: -43 % / -15 % / -17 %)
    assert rel["f16_all/ca0"] > rel["f16_all"] + 0.08 and abs(rel["f16_all/ca0"] - rel["bf16_all"]) < 0.10, rel


def test_train_loop_fp16_loss_scale_policy_without_queue_drain(tmp_path, monkeypatch):
    """train.train(compute_dtype="f16_all") from a loss scale that overflows fp16 (2^31): the loop's policy -- fed by the LAGGED copy of the scalars, no
    device->host sync per reporting interval -- must halve its way down until steps go through (the kernel skips overflowed steps meanwhile), count each
    overflow interval once, and end with finite parameters and a recorded validation loss."""
    from signaltrain_amd import train, audio, nn_proc
    nn_proc._QUIET = True
    made = []
    orig = nn_proc.st_model.engine

    def engine_with_big_scale(self, *a, **k):
        e = orig(self, *a, **k)
        if not made:
            e.loss_scale = 2.0 ** 31
        made.append(e)
        return e
    monkeypatch.setattr(nn_proc.st_model, "engine", engine_with_big_scale)
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        torch.manual_seed(0); np.random.seed(0)
        train.train(effect=audio.Compressor_4c(), epochs=1, n_data_points=16 * 600, batch_size=16, device=torch.device("cuda:0"),
                    num_workers=2, device_feed=True, compute_dtype="f16_all", lr_max=2e-4)
        eng = made[0]
        skipped = int(eng.scalars[5].item())
        assert 2.0 ** 12 <= eng.loss_scale < 2.0 ** 31, eng.loss_scale       # came down (one halving per two reporting intervals), did not collapse
        assert 0 < skipped < 500, skipped                                      # the early steps were skipped, the later ones ran
        assert eng.step_count == 600 and torch.isfinite(eng.params).all()
        lines = [l.split() for l in open("vl_avg_out.dat").read().strip().splitlines()]
        assert lines and np.isfinite(float(lines[-1][-1]))
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("dtype", ["f32", "f32x3"])
def test_graph_step_equals_eager_steps(golden_dir, dtype):
    """st_graph_*: the whole optimisation step captured once as a HIP graph (step counter and learning rate on the device, looked
    up in the device copy of the 1-cycle table as lr_sched[max(i-1, 0)], train.py:150) and replayed == the same steps launched
    eagerly with the host passing step and learning rate."""
    from signaltrain_amd.engine import StepEngine
    from signaltrain_amd import learningrate
    m, g, P, geo = _golden_model(golden_dir)
    gb = np.load(os.path.join(golden_dir, "g4b_backward_clip.npz"))
    d = m.engine(torch.zeros(3, 8192, device="cuda")).dims
    lrs, _ = learningrate.get_1cycle_schedule(lr_max=1e-3, n_data_points=30, epochs=1, batch_size=3)
    e1 = StepEngine(d, "cuda:0", compute_dtype=dtype); e1.load_state_dict(P)
    e2 = StepEngine(d, "cuda:0", compute_dtype=dtype); e2.load_state_dict(P)      # f32x3: the capture includes the plane kernels (> 64 KB of LDS: attribute set before the capture)
    kn = torch.from_numpy(gb["knobs"]).cuda()
    e2.graph_capture(3, lrs)
    for it in range(5):
        Xi = torch.from_numpy(np.roll(gb["x"], 23 * it, axis=1).copy()).cuda(); Yi = torch.from_numpy(np.roll(gb["y"], 23 * it, axis=1).copy()).cuda()
        e1.train_step(Xi, kn, Yi, float(lrs[max(it - 1, 0)]))
        e2.graph_step(Xi, kn, Yi)
        torch.cuda.synchronize()
        assert float(e2.scalars[6]) == it + 1 and abs(float(e2.scalars[7]) - np.float32(lrs[max(it - 1, 0)])) <= 1e-12
        assert torch.equal(e1.scalars[:5], e2.scalars[:5]), it
        assert (e1.params - e2.params).abs().max().item() <= 1e-7, it          # bias corrections: device double pow vs host double pow
    e2.graph_destroy()


def test_train_driver_file_dataset_three_knobs(tmp_path):
    """BASELINE configs[3] shape: train.train(datapath=...) on pre-recorded wav pairs with 3 knobs (audio.FileEffect +
    datasets.AudioFileDataSet, windows gathered on the device), K = 3 kernels, bf16 arithmetic; loss finite and decreasing-ish,
    the checkpoint records the file effect's knob names / ranges."""
    from tests.test_device_feed import make_file_dataset
    from signaltrain_amd import train, audio, nn_proc, misc, datasets
    nn_proc._QUIET = True
    root = make_file_dataset(str(tmp_path / "la2a"), n_train=4, n_val=2, seconds=1.0)
    fx = audio.FileEffect(root)
    ds = datasets.AudioFileDataSet(8192, fx, path=root + "/Train/", datapoints=64, y_size=2048, augment=True)
    x, y, k = ds.batch_device(16)
    assert x.shape == (16, 8192) and y.shape == (16, 2048) and k.shape == (16, 3) and x.is_cuda
    # a device item is a window of one of the files, target = last y_size samples of the same window (up to the polarity flip)
    xs = x[0].cpu().numpy(); ys = y[0].cpu().numpy(); found = False
    for a, b in zip(ds.x, ds.y):
        for sgn in (1.0, -1.0):
            idx = np.where(np.isclose(a[:len(a) - 8192], sgn * xs[0], atol=1e-7))[0]
            for p in idx:
                if np.allclose(a[p:p + 8192], sgn * xs, atol=1e-7) and np.allclose(b[p + 8192 - 2048:p + 8192], sgn * ys, atol=1e-7):
                    found = True
    assert found
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        model = train.train(effect=fx, epochs=2, n_data_points=256, batch_size=32, device=torch.device("cuda:0"), datapath=root,
                            device_feed=True, compute_dtype="bf16_all", lr_max=2e-4)
        assert model.num_knobs == 3
        sd, rv = misc.load_checkpoint("modelcheckpoint.tar", device="cpu")
        assert rv["knob_names"] == ['Limit/Comp', 'Gain', 'Gain Reduction'] and np.asarray(rv["knob_ranges"]).shape == (3, 2)
        assert sd["mpaec.aenc.fnn_addknobs.weight"].shape == (16, 19)
        vals = [float(l.split()[-1]) for l in open("vl_avg_out.dat").read().strip().splitlines()]
        assert len(vals) == 2 and all(np.isfinite(vals))
    finally:
        os.chdir(cwd)


def test_model_input_gradient_matches_torch_autograd():
    """Something trainable upstream of st_model gets its gradient: d loss / d x through the drop-in model (fused HIP backward +
    st_model_input_grad: half the conv-transpose of the analysis output gradient + the skip connection) against float64 torch autograd
    of the PyTorch-CPU restatement of the reference's op sequence (oracle/torch_cpu_step.py)."""
    from oracle import torch_cpu_step as TC
    from tests import gpu_checks as G
    from signaltrain_amd import nn_proc
    nn_proc._QUIET = True
    B, K = 3, 4
    geo, X, Y, KN, P = G.make_case(B, 13, K=K)
    m = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=K).cuda()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    x = torch.from_numpy(X).cuda().requires_grad_(True)
    y_hat, mag, mag_hat = m(x, torch.from_numpy(KN).cuda())
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    cy, cm, ch = (torch.randn(t.shape, device="cuda", generator=g) for t in (y_hat, mag, mag_hat))
    ((y_hat * cy).sum() + 1e-2 * (mag * cm).sum() + 1e-2 * (mag_hat * ch).sum()).backward()
    P64 = {k: torch.from_numpy(v).double() for k, v in P.items()}
    xd = torch.from_numpy(X).double().requires_grad_(True)
    yr, mr, hr = TC.forward(P64, xd, torch.from_numpy(KN).double())
    ((yr * cy.double().cpu()).sum() + 1e-2 * (mr * cm.double().cpu()).sum() + 1e-2 * (hr * ch.double().cpu()).sum()).backward()
    assert (y_hat.detach().double().cpu() - yr.detach()).abs().max() < 1e-4 * yr.detach().abs().max()
    e = (x.grad.double().cpu() - xd.grad).abs().max() / xd.grad.abs().max()
    assert e < 2e-5, float(e)


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_g14_edges_on_the_device(golden_dir, ci):
    """Golden G14 (the reference's forward / loss / fp32 autograd on three edge cases: a model WITHOUT knobs, one knob, digital silence -- tools/capture_golden_r6.py)
    against the HIP path directly: through the drop-in st_model + torch.autograd AND through the fused loss_backward entry.  y_hat and loss at 1e-4, the 36
    autoencoder gradients at 2e-4 of the tensor maximum, the STFT gradients' sampled rows / projections / L1 norms at the suite's tolerances (analysis bases:
    max(2e-4, 3 x the reference's own fp32 distance from float64))."""
    from tests.golden_util import g14_case, ae_keys, projections, SAMPLE_ROWS, STFT_KEYS
    from signaltrain_amd import nn_proc, loss_functions
    nn_proc._QUIET = True
    g = np.load(os.path.join(golden_dir, "g14_edges.npz"))
    geo, X, Y, KN, P, K = g14_case(ci)
    pre = f"c{ci}_"
    m = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=K)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.items()})
    m = m.to("cuda:0")
    x, kn, yt = torch.from_numpy(X).cuda(), torch.from_numpy(KN).cuda(), torch.from_numpy(Y).cuda()
    assert kn.shape == (X.shape[0], K)
    PROJ = projections(seed=23)
    F = geo["F"]

    def check(loss, y_hat, grads, what):
        assert abs(loss - float(g[pre + "loss"])) <= 1e-4 * abs(float(g[pre + "loss"])), (what, loss, float(g[pre + "loss"]))
        assert np.isfinite(y_hat).all() and np.abs(y_hat - g[pre + "y_hat"]).max() <= 1e-4 * np.abs(g[pre + "y_hat"]).max(), what
        for k in ae_keys():
            ref = g[pre + "g_" + k].astype(np.float64)
            got = grads[k].reshape(ref.shape)
            assert np.isfinite(got).all() and np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-12, (what, k)
        for k in STFT_KEYS:
            gk = grads[k].reshape(1024, 1024)
            tol = max(2e-4, 3.0 * float(g[pre + "ref_vs_f64_" + k]) if "analysis" in k else 2e-4) * float(g[pre + "max_" + k])
            assert np.isfinite(gk).all()
            assert np.abs(gk[SAMPLE_ROWS] - g[pre + "rows_" + k]).max() <= tol, (what, k)
            assert np.abs(PROJ @ gk - g[pre + "proj_" + k]).max() <= 1024 * tol, (what, k)
            assert abs(np.abs(gk).sum() - float(g[pre + "l1_" + k])) <= 1e-3 * float(g[pre + "l1_" + k]), (what, k)
    y, mag, mag_hat = m.forward(x, kn)
    sbf = torch.exp((7. / F) * torch.arange(0., F, device="cuda")).expand_as(mag_hat).float()
    loss = loss_functions.calc_loss(y, yt, mag_hat, scale_by_freq=sbf)
    loss.backward()
    check(loss.item(), y.detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in m.named_parameters()}, "autograd")
    eng = m.engine(x)
    outs = eng.loss_backward(x, kn, yt, want_outputs=True); torch.cuda.synchronize()
    check(float(eng.scalars[0]), outs[0].detach().cpu().numpy(), {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.layout.views(eng.grads).items()}, "fused")
    if ci == 2:                                  # the all-zero window: y_hat = 2 (syn + x / 2) of silence is what the model makes of a zero spectrum, the same bits as the oracle's convention allows
        assert np.isfinite(mag.detach().cpu().numpy()).all() and float(mag[0].detach().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", ["f32", "bf16_all", "f16_all"])
def test_model_without_knobs_every_engine_entry(dtype):
    """num_knobs = 0 (the reference builds it: nn_proc.py:92-93 concatenates an empty [B, 0] tensor; golden G14 case 0) through every engine entry that takes knobs:
    forward, loss_backward and a train step against the oracle (run_fused, the mode's tolerances), the graph step against the eager step (same bits), the exchange
    step with one rank, st_model_knob_grad (an empty [B, 0] result) and the standalone autoencoder module with return_acts.  An empty tensor has no data pointer: the C ABI
    takes knobs == NULL when K == 0 (it refused it until the third session of round 6, so st_model(num_knobs=0) could not run a forward) and hands the kernels, which issue
    one clamped and masked load of knobs[0] per row group, a resident address of its own."""
    from tests import gpu_checks as G
    from signaltrain_amd import nn_proc
    from signaltrain_amd.engine import StepEngine
    nn_proc._QUIET = True
    kw = dict(B=3, seed=41, K=0, steps=2)
    if dtype == "f32":
        res = G.run_fused(**kw)
    elif dtype == "bf16_all":
        with G.mixed_mode(2, half="bf16", tol_scale=G.mixed_mode.FUSED_TOL[2]): res = G.run_fused(**kw)
    else:
        with G.mixed_mode(2, half="f16", tol_scale=G.mixed_mode.FUSED_TOL_F16[2]): res = G.run_fused(**kw)
    bad = [r for r in res if not r["ok"]]
    assert not bad, [(r["name"], r["rel"], r["tol"]) for r in bad]
    geo, X, Y, KN, P = G.make_case(3, 41, K=0)
    assert KN.shape == (3, 0)
    d = G.dims_of(geo, 3, 0)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    a = StepEngine(d, G.DEV, compute_dtype=dtype); a.load_state_dict(P)
    b = StepEngine(d, G.DEV, compute_dtype=dtype); b.load_state_dict(P)
    c = StepEngine(d, G.DEV, compute_dtype=dtype); c.load_state_dict(P)
    lrs = [1e-3, 1e-3, 7e-4]
    b.graph_capture(3, lrs)
    for i in range(3):
        a.train_step(x, kn, y, lrs[max(i - 1, 0)])
        b.graph_step(x, kn, y)
        c.dp_train_step(x, kn, y, lrs[max(i - 1, 0)], force_exchange=False)
    torch.cuda.synchronize()
    assert torch.equal(a.params, b.params) and torch.equal(a.params, c.params) and bool(torch.isfinite(a.params).all())
    assert a.knob_grad(x, kn, torch.zeros_like(y)).shape == (3, 0)
    if dtype == "f32":
        ae = nn_proc.AsymAutoEncoder(T=geo["T"], R=64, K=0, OT=geo["OT"]).to(G.DEV)
        v = torch.rand(2, geo["T"], geo["F"], device=G.DEV)
        out, acts = ae.forward(v, torch.zeros(2, 0, device=G.DEV), skip_connections='sf', return_acts=True)
        assert out.shape == (2, geo["OT"], geo["F"]) and bool(torch.isfinite(out).all()) and len(acts) == 10


def test_growing_batch_recreates_the_engine_without_losing_state(golden_dir):
    """The reference's flows change the batch under a live model: train.py trains at batch_size, predict_long runs 200 windows at a time (predict_long.py:44-47), a last
    partial batch is smaller.  st_model sizes its engine for the largest batch seen so far and RE-CREATES it (parameters re-pointed into the new flat buffer) when a
    larger one arrives.  Model A trains on batches of 4, 9, 2 windows with a torch optimizer (the reference-style loop, train.py:131-151) -- the engine is rebuilt in
    the middle of the run, between a backward and the next forward; model B sees a 9-window batch first (no_grad), so it never rebuilds.  Same data, same kernels:
    parameters, gradients and Adam moments must agree bit for bit, and the fused engine entry must see the same parameters afterwards."""
    from signaltrain_amd import nn_proc, loss_functions
    nn_proc._QUIET = True
    mA, g, P, geo = _golden_model(golden_dir)
    mB, _, _, _ = _golden_model(golden_dir)
    rng = np.random.default_rng(12)
    X = torch.from_numpy((0.3 * rng.standard_normal((9, geo["L"]))).astype(np.float32)).cuda()
    Y = torch.from_numpy((0.3 * rng.standard_normal((9, geo["y"]))).astype(np.float32)).cuda()
    KN = torch.from_numpy((rng.random((9, 4)) - 0.5).astype(np.float32)).cuda()
    with torch.no_grad():
        mB.forward(X, KN)                                    # B's engine is sized for 9 windows from the start
    engB = mB.mpaec._engine
    oA = torch.optim.Adam(mA.parameters(), lr=1e-3); oB = torch.optim.Adam(mB.parameters(), lr=1e-3)
    rebuilt = []
    for nb in (4, 9, 2):
        for m, o in ((mA, oA), (mB, oB)):
            before = m.mpaec._engine
            y, mag, mag_hat = m.forward(X[:nb], KN[:nb])
            if m is mA:
                rebuilt.append(before is not None and m.mpaec._engine is not before)
            sbf = torch.exp((7. / 513) * torch.arange(0., 513, device="cuda")).expand_as(mag_hat).float()
            loss = loss_functions.calc_loss(y, Y[:nb], mag_hat, scale_by_freq=sbf)
            o.zero_grad(); loss.backward(); m.clip_grad_norm_(); o.step()
    assert rebuilt == [False, True, False] and mB.mpaec._engine is engB
    for (ka, pa), (kb, pb) in zip(mA.named_parameters(), mB.named_parameters()):
        assert ka == kb and torch.equal(pa, pb), ka
        assert torch.equal(pa.grad, pb.grad), ka
        sa, sb = oA.state[pa], oB.state[pb]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), ka
    # the fused entry of the rebuilt engine reads the very parameters the optimizer has been updating
    eA, eB = mA.engine(X[:2]), mB.engine(X[:2])
    eA.loss_backward(X[:2], KN[:2], Y[:2]); eB.loss_backward(X[:2], KN[:2], Y[:2]); torch.cuda.synchronize()
    assert torch.equal(eA.params, eB.params) and torch.equal(eA.grads, eB.grads) and float(eA.scalars[0]) == float(eB.scalars[0])


def test_train_driver_awkward_data_counts(tmp_path):
    """Ragged ends of the driver (train.py:167-263): a number of data points that is not a multiple of the batch (the remainder is dropped, like the reference's
    DataLoader with drop_last), a validation set smaller than one batch (n_data_points // 4 < batch_size: no validation minibatch -- the log files still get their line,
    the MAE column says nan), and less than one training minibatch, which the reference meets late with an UnboundLocalError (train.py:125-158) and this driver refuses
    up front by name."""
    from signaltrain_amd import train, audio, nn_proc
    nn_proc._QUIET = True
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        model = train.train(effect=audio.Compressor_4c(), epochs=2, n_data_points=100, batch_size=48, device=torch.device("cuda:0"))
        eng = model.engine(torch.zeros(48, 8192, device="cuda"))
        assert eng.step_count == 2 * (100 // 48)                          # 2 epochs x 2 whole minibatches, 4 windows dropped per epoch
        lines = open("val_err_mae.dat").read().split("\n")[:2]
        assert [l.split()[0] for l in lines] == ["1", "2"] and all(l.split()[1] == "nan" for l in lines)      # 25 validation windows < one batch of 48
        assert len(open("vl_avg_out.dat").read().strip().split("\n")) == 2
        assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
        with pytest.raises(ValueError, match="less than one minibatch"):
            train.train(effect=audio.Compressor_4c(), epochs=1, n_data_points=40, batch_size=48, device=torch.device("cuda:0"))
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("n", [1000, 7000, 8192, 8193, 8192 + 2048, 8192 + 5 * 2048 - 1])
def test_predict_long_ragged_lengths(golden_dir, n):
    """predict_long (utils/predict_long.py:30-79) at the ragged ends: a signal shorter than one window (the reference's sliding_window builds a negative window count
    there, audio.py:41-47; here the signal is zero-padded to one window and what is returned is what has a full lookback: nothing below chunk - out_chunk samples),
    exactly one window, one sample more, whole hops, one sample short of whole hops.  Length and values against the oracle run window by window."""
    from oracle import st_oracle as O
    from signaltrain_amd.predict import predict_long
    m, g, P, geo = _golden_model(golden_dir)
    L, ysz = geo["L"], geo["y"]
    rng = np.random.default_rng(n)
    sig = (0.4 * np.sin(np.arange(n) * 0.013) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    kn = np.array([0.1, -0.3, 0.25, -0.45], np.float32)
    y = predict_long(sig, kn, m, L, ysz, batch_size=2)
    want = max(n - (L - ysz), 0)
    assert y.dtype == np.float32 and y.shape == (want,)
    if want == 0:
        return
    step = ysz
    pad = (L - n) if n < L else ((step - (n - L) % step) % step)
    sp = np.concatenate([sig, np.zeros(pad, np.float32)])
    xw = np.stack([sp[i:i + L] for i in range(0, sp.size - L + 1, step)])
    ref = O.model_fwd(np.ascontiguousarray(xw), np.tile(kn, (xw.shape[0], 1)), P, geo)[0].reshape(-1)[:want]
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max()


def test_switching_arithmetic_on_a_live_engine():
    """StepEngine.set_arithmetic / st_model.set_compute_dtype on a LIVE engine: the workspace depends on the arithmetic level (fp32 autoencoder layers keep their activations
    for the backward -- 73 MB at B = 64 --, the 16-bit levels carry operand copies).  An engine created from dims that carry a 16-bit level and then switched to fp32 used to
    own a workspace sized for the 16-bit level (the fp32 backward would have written past its end); it is sized for the largest level now.  After the switch the engine's
    step is bit for bit the step of an engine that was fp32 from the start, and switching back reproduces the 16-bit step."""
    from tests import gpu_checks as G
    from signaltrain_amd import _lib
    from signaltrain_amd.engine import StepEngine
    geo, X, Y, KN, P = G.make_case(8, 19, K=4)
    B = 64
    rng = np.random.default_rng(2)
    X = (np.tile(X, (B // 8, 1)) * rng.uniform(0.5, 1.0, (B, 1))).astype(np.float32); Y = np.tile(Y, (B // 8, 1)).astype(np.float32); KN = np.tile(KN, (B // 8, 1)).astype(np.float32)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    d16 = G.dims_of(geo, B, 4).with_arith(prec=_lib.PREC["bf16_all"])
    live = StepEngine(d16, G.DEV, compute_dtype="bf16_all"); live.load_state_dict(P)
    assert live.ws.numel() >= int(live.lib.st_workspace_bytes(__import__("ctypes").byref(d16.with_arith(prec=_lib.PREC["f32"]))))
    ref16 = StepEngine(G.dims_of(geo, B, 4), G.DEV, compute_dtype="bf16_all"); ref16.load_state_dict(P)
    ref32 = StepEngine(G.dims_of(geo, B, 4), G.DEV, compute_dtype="f32"); ref32.load_state_dict(P)
    guard = torch.full((1 << 20,), 7.0, device=G.DEV)                        # something allocated right after the engines: must stay untouched
    live.train_step(x, kn, y, 1e-3); ref16.train_step(x, kn, y, 1e-3)
    assert torch.equal(live.params, ref16.params)
    live.set_arithmetic("f32"); live.load_state_dict(P); live.m.zero_(); live.v.zero_(); live.step_count = 0
    live.train_step(x, kn, y, 1e-3); ref32.train_step(x, kn, y, 1e-3)
    torch.cuda.synchronize()
    assert torch.equal(live.params, ref32.params) and torch.equal(live.grads, ref32.grads) and float(live.scalars[0]) == float(ref32.scalars[0])
    live.set_arithmetic("bf16_all"); live.load_state_dict(P); live.m.zero_(); live.v.zero_(); live.step_count = 0
    ref16.load_state_dict(P); ref16.m.zero_(); ref16.v.zero_(); ref16.step_count = 0
    live.train_step(x, kn, y, 1e-3); ref16.train_step(x, kn, y, 1e-3)
    torch.cuda.synchronize()
    assert torch.equal(live.params, ref16.params) and bool((guard == 7.0).all())


def test_engine_sized_for_a_large_batch_runs_a_smaller_one_that_needs_more_workspace():
    """st_workspace_bytes is not monotonic in the batch (tests/test_abi_and_host.py::test_one_workspace_serves_every_smaller_batch): at the default geometry 585 windows need
    85.6 MB more than 586.  An engine created for up to 600 windows must run 585 inside its own workspace: its capacity is checked BEFORE anything is launched, a guard
    allocation behind it stays untouched, and the step is bit for bit that of an engine created for exactly 585 windows."""
    import ctypes as C
    from tests import gpu_checks as G
    from signaltrain_amd.engine import StepEngine
    geo, X, Y, KN, P = G.make_case(5, 23, K=4)
    B = 585
    rng = np.random.default_rng(4)
    X = (np.tile(X, (B // 5, 1)) * rng.uniform(0.5, 1.0, (B, 1))).astype(np.float32); Y = np.tile(Y, (B // 5, 1)).astype(np.float32); KN = np.tile(KN, (B // 5, 1)).astype(np.float32)
    big = StepEngine(G.dims_of(geo, 600, 4), G.DEV, max_batch=600); big.load_state_dict(P)
    need = int(big.lib.st_workspace_bytes(C.byref(G.dims_of(geo, B, 4))))
    assert need > int(big.lib.st_workspace_bytes(C.byref(G.dims_of(geo, 600, 4))))          # the non-monotonic spot itself
    assert big.ws.numel() >= need                                                              # ... which the engine covers (never launch into a short workspace)
    guard = torch.full((1 << 22,), 3.0, device=G.DEV)
    exact = StepEngine(G.dims_of(geo, B, 4), G.DEV); exact.load_state_dict(P)
    x, kn, y = G.t(X), G.t(KN), G.t(Y)
    big.train_step(x, kn, y, 1e-3); exact.train_step(x, kn, y, 1e-3)
    torch.cuda.synchronize()
    assert torch.equal(big.params, exact.params) and torch.equal(big.grads, exact.grads) and float(big.scalars[0]) == float(exact.scalars[0])
    assert bool((guard == 3.0).all()) and bool(torch.isfinite(big.params).all())
