"""Loss functions with the reference's signatures (signaltrain/loss_functions.py).

These torch-op versions exist for API compatibility (validation, custom training loops through autograd).
The fused training step (engine.StepEngine.train_step) computes calc_loss's default branch
(log-cosh + frequency-weighted L1, loss_functions.py:36) and its gradient inside the HIP kernels."""
import torch


def logcosh(y_hat, y):
    """loss_functions.py:9-10, evaluated overflow-free: log cosh d = |d| + log1p(exp(-2|d|)) - log 2."""
    d = (y - y_hat).abs()
    return torch.mean(d + torch.log1p(torch.exp(-2.0 * d)) - 0.6931471805599453)


def mse(x, x_hat):
    return torch.mean((x - x_hat) ** 2)


def mae(x, x_hat):
    return torch.mean(torch.abs(x - x_hat))


def calc_loss(y_hat, y_cuda, mag_hat, batch_size=20, scale_by_freq=None, l1_lambda=2e-5, reg_logcosh=False):
    """loss_functions.py:26-43."""
    if not reg_logcosh:
        if scale_by_freq is None:
            return logcosh(y_hat, y_cuda) + l1_lambda * torch.abs(mag_hat).mean()
        return logcosh(y_hat, y_cuda) + l1_lambda / 10 * torch.abs(mag_hat * scale_by_freq).mean()
    lc = lambda t: t.abs() + torch.log1p(torch.exp(-2.0 * t.abs())) - 0.6931471805599453
    if scale_by_freq is None:
        return logcosh(y_hat, y_cuda) + l1_lambda * torch.mean(lc(mag_hat))
    return logcosh(y_hat, y_cuda) + l1_lambda / 10 * torch.mean(scale_by_freq * lc(mag_hat))
