"""Checkpoint I/O in the reference's on-disk format (signaltrain/misc.py:21-66).

The FORMAT is the contract: `modelcheckpoint.tar` is a `torch.save`d dict with the keys of CKPT_KEYS; its
'state_dict' holds the 40 tensors of st_model under the reference's names and its 'optimizer' entry is a
`torch.optim.Adam.state_dict()` (per-parameter 'step' / 'exp_avg' / 'exp_avg_sq' keyed by parameter index + one param
group).  Files written by the reference load here and vice versa (tests/golden/g10_checkpoint.npz pins the layout from a
reference-written file).  Beyond the reference: the optimizer state is actually RESTORED on resume (the reference saves it
but never reads it back, train.py:229) -- `flatten_optimizer_state` / `adam_state_dict` convert between torch's layout and
the engine's flat moment buffers.
"""
import os
import sys
import numpy as np
import torch

CKPT_KEYS = ('epoch', 'state_dict', 'optimizer', 'effect_name', 'knob_names', 'knob_ranges', 'scale_factor',
             'shrink_factor', 'in_chunk_size', 'out_chunk_size', 'sr')
# run values assumed for old checkpoints that lack them (misc.py:51-57)
LEGACY_RUN_VALUES = {'sr': 44100, 'scale_factor': 1, 'shrink_factor': 4, 'in_chunk_size': 8192, 'out_chunk_size': 2048,
                     'knob_names': ['thresh', 'ratio', 'attackTime', 'releaseTime'],
                     'knob_ranges': np.array([[-30, 0], [1, 5], [1e-3, 4e-2], [1e-3, 4e-2]])}


def save_checkpoint(checkpointname, model, epoch, parallel, optimizer, effect, sr):
    """misc.py:21-35.  `optimizer`: a torch optimizer, or anything whose state_dict() is in torch.optim.Adam's layout
    (engine.StepEngine.optimizer_view)."""
    print(f'\nsaving model to {checkpointname}', end="")
    net = model.module if parallel else model
    run = {'epoch': epoch + 1, 'effect_name': effect.name, 'knob_names': effect.knob_names, 'knob_ranges': effect.knob_ranges,
           'scale_factor': net.scale_factor, 'shrink_factor': net.shrink_factor, 'in_chunk_size': net.in_chunk_size,
           'out_chunk_size': net.out_chunk_size, 'sr': sr}
    payload = {'state_dict': net.state_dict(), 'optimizer': optimizer.state_dict() if optimizer is not None else {}}
    torch.save({k: (run[k] if k in run else payload[k]) for k in CKPT_KEYS}, checkpointname)


def load_checkpoint(checkpointname, fatal=False, device="cuda"):
    """misc.py:38-66: (state_dict, run_values); every non-state_dict entry of the file is a run value, old files get the
    legacy defaults.  weights_only=False: the file holds a numpy array (knob_ranges)."""
    if not os.path.isfile(checkpointname):
        if fatal:
            print("Error, no checkpoint found")
            sys.exit(1)
        return {}, {}
    print("\n***** Checkpoint file found. Loading weights.")
    blob = torch.load(checkpointname, map_location=device, weights_only=False)
    rv = dict(LEGACY_RUN_VALUES)
    rv.update({k: v for k, v in blob.items() if 'state_dict' not in k})
    return blob['state_dict'], rv


# ---------------------------------------------------------------------------------------- optimizer state <-> flat buffers
def flatten_optimizer_state(opt_sd, shapes):
    """torch.optim.Adam.state_dict() -> {'step': int, 'lr': float, 'betas', 'eps', 'exp_avg': [flat np.float32 per
    parameter], 'exp_avg_sq': [...]}; None if `opt_sd` is not in that layout (e.g. the round-1 stand-in dict or {}).
    `shapes`: the parameter shapes in state_dict order, used to validate."""
    try:
        groups, state = opt_sd['param_groups'], opt_sd['state']
        order = [i for g in groups for i in g['params']]
        if len(order) != len(shapes) or not all(i in state for i in order):
            return None
        out = {'lr': float(groups[0]['lr']), 'betas': tuple(groups[0]['betas']), 'eps': float(groups[0]['eps']),
               'exp_avg': [], 'exp_avg_sq': []}
        steps = set()
        for i, shp in zip(order, shapes):
            st = state[i]
            m, v = st['exp_avg'], st['exp_avg_sq']
            if tuple(m.shape) != tuple(shp) or tuple(v.shape) != tuple(shp):
                return None
            out['exp_avg'].append(np.ascontiguousarray(m.detach().cpu().numpy(), dtype=np.float32).ravel())
            out['exp_avg_sq'].append(np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32).ravel())
            steps.add(int(float(st['step'])))
        if len(steps) != 1:
            return None
        out['step'] = steps.pop()
        return out
    except (KeyError, TypeError, AttributeError, IndexError):
        return None


def adam_state_dict(step, lr, exp_avg, exp_avg_sq, betas=(0.9, 0.999), eps=1e-8, extra_group=None):
    """The inverse: per-parameter tensors (lists in state_dict order) -> torch.optim.Adam.state_dict() layout, loadable by
    `torch.optim.Adam(model.parameters()).load_state_dict`."""
    n = len(exp_avg)
    state = {i: {'step': torch.tensor(float(step)), 'exp_avg': exp_avg[i], 'exp_avg_sq': exp_avg_sq[i]} for i in range(n)}
    group = {'lr': float(lr), 'betas': tuple(betas), 'eps': float(eps), 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
             'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'decoupled_weight_decay': False,
             'params': list(range(n))}
    if extra_group:
        group.update(extra_group)
    return {'state': state, 'param_groups': [group]}
