"""CPU baseline ("port") for bench.py's cpu_baseline leg -- TEST/BENCH INFRASTRUCTURE ONLY.

The reference's op sequence restated with the same PyTorch calls it makes (cls_fe_dft.py:50-58,102-115:
F.conv1d / F.conv_transpose1d; nn_proc.py:77-126: F.linear + F.elu; nn_proc.py:305-340; loss_functions.py:26-36;
train.py:131-151: backward, L1 clip of the STFT grads, torch.optim.Adam), so that its CPU timing stands in for
the reference's own CPU path on the GPU box, where /root/reference does not exist.  Validated against the
numpy oracle in tests/test_abi_and_host.py::test_cpu_port_matches_oracle."""
import numpy as np
import torch
import torch.nn.functional as F

AE_LAYERS = ("fnn_enc", "fnn_enc2", "fnn_enc3", "fnn_enc4", "fnn_addknobs", "fnn_dec4", "fnn_dec3", "fnn_dec2", "fnn_dec")
STFT_KEYS = ("mpaec.dft_analysis.conv_analysis_real.weight", "mpaec.dft_analysis.conv_analysis_imag.weight",
             "mpaec.dft_synthesis.conv_synthesis_real.weight", "mpaec.dft_synthesis.conv_synthesis_imag.weight")


def _ae(P, pref, v, knobs, mode):
    xi = v.transpose(2, 1)
    z = xi
    for i, n in enumerate(AE_LAYERS[:-1]):
        if n == "fnn_addknobs":
            z = torch.cat((z, knobs.unsqueeze(1).repeat(1, z.size(1), 1)), 2)
        z = F.elu(F.linear(z, P[f"{pref}.{n}.weight"], P[f"{pref}.{n}.bias"]))
    out = F.elu(F.linear(z, P[f"{pref}.fnn_dec.weight"], P[f"{pref}.fnn_dec.bias"]))
    if mode == 'sf':
        out = out * xi[:, :, -out.shape[2]:]
    return out.transpose(2, 1)


def forward(P, x, knobs, N=1024, H=384):
    half = N // 2 + 1
    w = (x / 2).view(x.shape[0], 1, -1)
    re = F.conv1d(w, P[STFT_KEYS[0]], stride=H, padding=N).transpose(1, 2)[:, :, :half]
    im = F.conv1d(w, P[STFT_KEYS[1]], stride=H, padding=N).transpose(1, 2)[:, :, :half]
    mag = torch.norm(torch.cat((re.unsqueeze(0), im.unsqueeze(0)), 0), 2, dim=0)
    phs = torch.atan2(im, re + 1e-7)
    mag_hat = _ae(P, "mpaec.aenc", mag, knobs, 'sf')
    phs_hat = _ae(P, "mpaec.phs_aenc", phs, knobs, '')
    phs_hat = phs_hat + phs[:, -phs_hat.shape[1]:, :]
    ar, ai = (mag_hat * torch.cos(phs_hat)).transpose(1, 2), (mag_hat * torch.sin(phs_hat)).transpose(1, 2)
    ar = torch.cat((ar, ar[:, 1:-1, :].flip(1)), 1)
    ai = torch.cat((ai, (-ai[:, 1:-1, :]).flip(1)), 1)
    wave = F.conv_transpose1d(ar, P[STFT_KEYS[2]], stride=H) + F.conv_transpose1d(ai, P[STFT_KEYS[3]], stride=H)
    syn = wave[:, 0, N:-N]
    return 2 * (syn + x[:, -syn.shape[-1]:] / 2), mag, mag_hat


class CpuPort:
    def __init__(self, params_np, lr=1e-4):
        self.P = {k: torch.from_numpy(np.array(v, dtype=np.float32)).requires_grad_(True) for k, v in params_np.items()}
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr, weight_decay=0)
        self.sbf = None

    def step(self, x, knobs, y, lr=None):
        if lr is not None:
            self.opt.param_groups[0]['lr'] = lr
        y_hat, mag, mag_hat = forward(self.P, x, knobs)
        if self.sbf is None or self.sbf.shape != mag_hat.shape:
            Fb = mag_hat.shape[-1]
            self.sbf = torch.exp((7. / Fb) * torch.arange(0., Fb)).expand_as(mag_hat).float()
        loss = torch.mean(torch.log(torch.cosh(y - y_hat))) + 2e-5 / 10 * torch.abs(mag_hat * self.sbf).mean()
        self.opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([self.P[k] for k in STFT_KEYS], max_norm=1., norm_type=1)
        self.opt.step()
        return float(loss.item())
