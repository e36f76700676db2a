"""Checkpoint I/O in the reference's format (signaltrain/misc.py:21-66): a torch.save'd dict
{'epoch','state_dict','optimizer','effect_name','knob_names','knob_ranges','scale_factor','shrink_factor',
 'in_chunk_size','out_chunk_size','sr'} written to modelcheckpoint.tar."""
import os
import sys
import numpy as np
import torch


def save_checkpoint(checkpointname, model, epoch, parallel, optimizer, effect, sr):
    """misc.py:21-35.  `optimizer` may be a torch optimizer or anything with state_dict()."""
    print(f'\nsaving model to {checkpointname}', end="")
    model2save = model.module if parallel else model
    state = {'epoch': epoch + 1, 'state_dict': model2save.state_dict(),
             'optimizer': optimizer.state_dict() if optimizer is not None else {},
             'effect_name': effect.name, 'knob_names': effect.knob_names, 'knob_ranges': effect.knob_ranges,
             'scale_factor': model2save.scale_factor, 'shrink_factor': model2save.shrink_factor,
             'in_chunk_size': model2save.in_chunk_size, 'out_chunk_size': model2save.out_chunk_size, 'sr': sr}
    torch.save(state, checkpointname)


def load_checkpoint(checkpointname, fatal=False, device="cuda"):
    """misc.py:38-66: returns (state_dict, run_values); back-compat defaults for old files."""
    state_dict, rv = {}, {}
    if os.path.isfile(checkpointname):
        print("\n***** Checkpoint file found. Loading weights.")
        checkpoint = torch.load(checkpointname, map_location=device, weights_only=False)   # holds a numpy array (knob_ranges)
        state_dict = checkpoint['state_dict']
        rv.setdefault('sr', 44100)
        rv.setdefault('scale_factor', 1)
        rv.setdefault('shrink_factor', 4)
        rv.setdefault('in_chunk_size', 8192)
        rv.setdefault('out_chunk_size', 2048)
        rv.setdefault('knob_names', ['thresh', 'ratio', 'attackTime', 'releaseTime'])
        rv.setdefault('knob_ranges', np.array([[-30, 0], [1, 5], [1e-3, 4e-2], [1e-3, 4e-2]]))
        for key, value in checkpoint.items():
            if 'state_dict' not in key:
                rv[key] = value
    elif fatal:
        print("Error, no checkpoint found")
        sys.exit(1)
    return state_dict, rv
