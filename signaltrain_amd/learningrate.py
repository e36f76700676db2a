"""1-cycle learning-rate / momentum tables -- mirror of signaltrain/learningrate.py:14-52 (host side, numpy)."""
import numpy as np


def get_1cycle_schedule(lr_max=1e-3, n_data_points=8000, epochs=200, batch_size=40):
    """Look-up tables (lrs, moms) of length n_data_points*epochs//batch_size: cosine ramp from
    lr_max/15 up over the first 30 % of iterations, cosine anneal down to lr_max/1500."""
    pct_start, div_factor = 0.3, 15.0
    lr_start = lr_max / div_factor
    lr_end = lr_start / 1e2
    n_iter = n_data_points * epochs // batch_size
    a1 = int(n_iter * pct_start)
    a2 = n_iter - a1
    lrs = np.concatenate(((lr_max - lr_start) * (1 - np.cos(np.linspace(0, np.pi, a1))) / 2 + lr_start,
                          (lr_max - lr_end) * (1 + np.cos(np.linspace(0, np.pi, a2))) / 2 + lr_end))
    mom_min, mom_max = 0.85, 0.95
    mom_avg, mom_amp = (mom_min + mom_max) / 2, (mom_max - mom_min) / 2
    moms = np.concatenate((mom_avg + mom_amp * np.cos(np.linspace(0, np.pi, a1)),
                           mom_avg - mom_amp * np.cos(np.linspace(0, np.pi, a2))))
    return lrs, moms
