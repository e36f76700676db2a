"""Long-file inference -- mirror of utils/predict_long.py:30-79 (predict_long).  (The plotting helper calc_ct, :82-99, runs the
CPU effect chunk by chunk and is not part of the accelerated path: it is not mirrored.)

The reference windows the signal on the host (audio.sliding_window, audio.py:23-49), ships every batch of
overlapping windows to the device and appends the outputs on the host.  Here the (zero-padded) signal is uploaded
ONCE; the overlapping windows are a strided device view (`unfold`), each batch goes through the HIP forward
(st_model.forward / StepEngine.forward) and the non-overlapping outputs are written straight into one device buffer.
Same arguments, same return value: a 1-D float32 numpy array of len(signal) - (chunk_size - out_chunk_size) samples
(the first window's lookback has no prediction, exactly as in the reference)."""
import numpy as np
import torch



def predict_long(signal, knobs_nn, model, chunk_size, out_chunk_size, sr=44100, effect=None, device=None, compand=False,
                 batch_size=200, verbose=False):
    device = torch.device(device) if device is not None else next(model.parameters()).device
    signal = np.ascontiguousarray(signal, dtype=np.float32)
    if compand:                                                        # predict_long.py:38-40 compands every window; the map is elementwise, so the signal once
        from . import audio
        print("Companding input")
        signal = np.ascontiguousarray(audio.mu_compand(signal), dtype=np.float32)
    overlap = chunk_size - out_chunk_size
    step = chunk_size - overlap                                        # == out_chunk_size
    # audio.sliding_window's padding rule (audio.py:42-45): zeros at the end until the windows tile the signal
    n = signal.shape[-1]
    if n < chunk_size:
        pad = chunk_size - n
    else:
        rem = (n - chunk_size) % step
        pad = (step - rem) if rem != 0 else 0
    sig = torch.zeros(n + pad, dtype=torch.float32, device=device)
    sig[:n] = torch.from_numpy(signal).to(device)
    x = sig.unfold(0, chunk_size, step)                                # [nwin, chunk_size] view, no copy
    nwin = x.shape[0]
    if verbose:
        print("predict_long: chunk_size, out_chunk_size, overlap = ", chunk_size, out_chunk_size, overlap)
        print("predict_long: x.shape, signal.shape = ", tuple(x.shape), signal.shape)
    kn_row = torch.as_tensor(np.asarray(knobs_nn, dtype=np.float32).reshape(1, -1), device=device)
    y_pred = torch.empty(nwin * out_chunk_size, dtype=torch.float32, device=device)
    bs = min(int(batch_size), nwin)
    bmax = max(int(np.round(nwin / bs)), 1)                            # the reference's batching rule (predict_long.py:53)
    with torch.no_grad():
        for b in range(bmax):
            bstart = b * bs
            nb = (nwin - bstart) if b == bmax - 1 else bs              # the last batch takes whatever is left
            xb = x[bstart:bstart + nb].contiguous()
            y_hat = model.forward(xb, kn_row.expand(nb, -1).contiguous())[0]
            y_pred[bstart * out_chunk_size:(bstart + nb) * out_chunk_size] = y_hat.reshape(-1)
    unique = chunk_size + (nwin - 1) * (chunk_size - overlap)          # predict_long.py:72-73
    num_extra = unique - n
    out = y_pred.cpu().numpy()
    return out[0:-num_extra] if num_extra > 0 else out

