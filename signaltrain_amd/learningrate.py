"""1-cycle learning-rate and momentum tables (host side, float64) with the values of signaltrain/learningrate.py:14-52.

Built from the schedule's definition rather than from the reference's code: one cycle of `n = n_data_points*epochs//batch_size`
iterations made of two half-cosine segments,

    warm-up   (first  floor(0.3 n) iterations):  lr rises  lr_max/15  ->  lr_max,  momentum falls 0.95 -> 0.85
    annealing (remaining iterations):            lr falls  lr_max     ->  lr_max/1500, momentum rises 0.85 -> 0.95

each segment sampled at `linspace(0, pi, len)` (both end points included, so lr_max appears twice in a row at the joint).
tests/golden/g6_1cycle.npz (captured from the reference) pins the table bit for bit.  Adam ignores the momentum table
(train.py:151 writes it into a key Adam never reads); it is returned because the reference returns it.
"""
import numpy as np

WARMUP_FRACTION = 0.3
START_DIVISOR = 15.0        # lr(0)  = lr_max / 15
END_DIVISOR = 100.0         # lr(-1) = lr(0) / 100
MOMENTUM_RANGE = (0.85, 0.95)


def _ramp(lo, hi, n, rising):
    """n samples of a half-cosine between lo and hi over theta = linspace(0, pi, n): lo + (hi - lo) (1 -/+ cos theta) / 2,
    i.e. lo -> hi when rising, hi -> lo when not.  (This association of the operations is the one the golden table pins.)"""
    c = np.cos(np.linspace(0.0, np.pi, n))
    return (hi - lo) * ((1.0 - c) if rising else (1.0 + c)) / 2.0 + lo


def _swing(lo, hi, n, falling_first):
    """Momentum segment: midpoint +/- half-range x cos(theta)."""
    mid, half = (lo + hi) / 2.0, (hi - lo) / 2.0
    c = np.cos(np.linspace(0.0, np.pi, n))
    return mid + half * c if falling_first else mid - half * c


def get_1cycle_schedule(lr_max=1e-3, n_data_points=8000, epochs=200, batch_size=40):
    """(lrs, moms): float64 look-up tables indexed by the iteration number."""
    n = n_data_points * epochs // batch_size
    n_up = int(n * WARMUP_FRACTION)
    n_down = n - n_up
    lr_first = lr_max / START_DIVISOR
    lr_last = lr_first / END_DIVISOR
    mom_lo, mom_hi = MOMENTUM_RANGE
    lrs = np.concatenate((_ramp(lr_first, lr_max, n_up, rising=True), _ramp(lr_last, lr_max, n_down, rising=False)))
    moms = np.concatenate((_swing(mom_lo, mom_hi, n_up, falling_first=True), _swing(mom_lo, mom_hi, n_down, falling_first=False)))
    return lrs, moms
