"""Training driver -- mirror of signaltrain/train.py (train :167-278, train_loop :84-164, eval_status_save :28-80).

Same call signature, same step ordering (forward -> calc_loss -> backward -> L1 clip of the STFT gradients ->
Adam -> learning-rate write taking effect on the NEXT step, train.py:104-151), same log files
(vl_avg_out.dat, val_err_mae.dat) and checkpoint format.  The step itself is one fused call into
libsignaltrain_hip.so (engine.StepEngine.train_step); with torch.distributed initialised (one process per GPU)
each rank trains on its shard and dp.DataParallel all-reduces the flat gradient over RCCL.
Plots (io_methods.plot_valdata / plot_spectrograms) are out of scope and skipped.
"""
import time
import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from . import audio, datasets, learningrate, loss_functions, misc, nn_proc
from .dp import DataParallel


def eval_status_save(model, engine, effect, epoch, epochs, lr, mom, device, dataloader_val, logfilename, first_time,
                     beta, vl_avg, out_checkpointname, parallel, optimizer, data_point, smoothed_loss, y_size, sr,
                     status_every, plot_every=10, cp_every=25, scale_by_freq=None, is_main=True):
    """train.py:28-80: validation pass, log files, checkpoint cadence."""
    val_batch_num, val_mae = 0, float("nan")
    for x_val, y_val, knobs_val in dataloader_val:
        val_batch_num += 1
        x_c, y_c, k_c = x_val.to(device), y_val.to(device).float(), knobs_val.to(device)
        y_hat, mag_val, mag_val_hat = engine.forward(x_c, k_c)
        if scale_by_freq is None or scale_by_freq.shape != mag_val_hat.shape:
            F = mag_val_hat.shape[-1]
            scale_by_freq = torch.exp((7. / F) * torch.arange(0., F, device=device)).expand_as(mag_val_hat).float()
        loss_val = loss_functions.calc_loss(y_hat, y_c, mag_val_hat, scale_by_freq=scale_by_freq)
        vl_avg = beta * vl_avg + (1 - beta) * loss_val.item()
        val_mae = loss_functions.mae(y_hat, y_c).item()
        if 0 == val_batch_num % status_every and is_main:
            print(f"\repoch {epoch+1}/{epochs}, time: {time.time()-first_time:.2f}: lr={lr:.2e},mom={mom:.3f} data_point {data_point}: "
                  f"loss: {smoothed_loss:.3e} val_loss: {vl_avg:.3e}   ", end="")
    if is_main:
        with open(logfilename, "a") as f:
            f.write(f"{epoch+1} {vl_avg:.3e}\n")
        with open("val_err_mae.dat", "a") as f:
            f.write(f"{epoch+1} {val_mae:.3e}\n")
        if ((epoch + 1) % cp_every == 0) or (epoch == epochs - 1):
            misc.save_checkpoint(out_checkpointname, model, epoch, parallel, optimizer, effect, sr)
        if (epoch + 1) == 1:
            hours = (time.time() - first_time) * (epochs - 1) / 3600.0
            print(f"\nExpect run to finish in roughly {hours:.1f} hours")
    return vl_avg


def train_loop(model, engine, effect, device, epochs, batch_size, lr_sched, mom_sched, dataloader, dataloader_val,
               y_size, logfilename, out_checkpointname, plot_every=10, cp_every=25, sr=44100, lr_max=1e-4,
               start_epoch=0, start_iter=0, lr_resume=None):
    """train.py:84-164.  start_epoch / start_iter / lr_resume: position restored from a checkpoint (epoch counter, optimizer
    step count = position in the 1-cycle table, the learning rate that sat in the optimizer) -- the reference restarts all
    three at zero on resume (train.py:229 TODO)."""
    dp = DataParallel(engine)
    try:
        return _train_loop(dp, model, engine, effect, device, epochs, batch_size, lr_sched, mom_sched, dataloader, dataloader_val,
                           y_size, logfilename, out_checkpointname, plot_every, cp_every, sr, lr_max, start_epoch, start_iter, lr_resume)
    finally:
        dp.close()          # the library-owned RCCL communicator, its stream and events (before the caller destroys the process group)


def seed_data_streams(rank):
    """Fold the rank into the DATA random streams (numpy's global state, torch's global generator) -- after the model has been
    initialised from the common seed (run_train.py:20-21 seeds 218 on every rank): each rank must draw different minibatches, or the
    all-reduced gradient is just the single-GPU gradient N times over.  (The reference's CPU workers re-seed from OS entropy,
    datasets.py:54-61; the device feeds draw from these two streams.)  Rank 0 keeps its streams."""
    if rank:
        np.random.seed((int(np.random.randint(0, 2 ** 31 - 1)) + 7919 * int(rank)) % (2 ** 31 - 1))
        torch.manual_seed(int(torch.initial_seed()) + 7919 * int(rank))


def _train_loop(dp, model, engine, effect, device, epochs, batch_size, lr_sched, mom_sched, dataloader, dataloader_val,
                y_size, logfilename, out_checkpointname, plot_every, cp_every, sr, lr_max, start_epoch, start_iter, lr_resume):
    dp.broadcast_parameters()
    is_main = (not dist.is_initialized()) or dist.get_rank() == 0
    iter_count, batch_num, status_every = int(start_iter), 0, 10
    avg_loss, vl_avg, beta = 0.0, 0.0, 0.98
    smoothed_loss = float("nan")          # the reference leaves this unbound for epochs shorter than 10 batches
    first_time = time.time()
    lr_in_optimizer = lr_sched[0] if lr_resume is None else lr_resume      # torch.optim.Adam(lr=lr_sched[0]), train.py:228
    engine.lr = float(lr_in_optimizer)
    opt = engine.optimizer_view()         # state_dict() in torch.optim.Adam's layout for misc.save_checkpoint
    windows, t_train, t_last = 0, 0.0, 0.0
    clean_steps, n_loss = 0, 0
    seen_overflows, skip_overflow_report = 0, False
    engine.scalars[5] = 0.0               # steps skipped on overflow, cumulative from here (device-side store, no sync)
    for epoch in range(int(start_epoch), epochs):
        if is_main:
            print("")
        data_point = 0
        t_ep = time.time()
        for x, y, knobs in dataloader:
            if x.shape[0] != batch_size:  # keep batches uniform (the reference's expand_as quirk)
                continue
            x_c, y_c, k_c = x.to(device), y.to(device).float(), knobs.to(device)
            lr = lr_sched[min(iter_count, len(lr_sched) - 1)]
            mom = mom_sched[min(iter_count, len(mom_sched) - 1)]
            data_point += batch_size
            dp.train_step(x_c, k_c, y_c, lr_in_optimizer)          # forward, loss, backward, clip, Adam (train.py:112-147)
            batch_num += 1
            if 0 == batch_num % status_every:                        # train.py:124-129 (the only device->host sync)
                # the progress line shows the loss of the PREVIOUS reporting point (no queue drain, dp.mean_loss_lagged); f16 needs the
                # overflow counter now anyway, and data parallel its collective
                f16 = engine.compute_dtype.startswith("f16")
                lagged = dp.world == 1                               # one process: no queue drain, also not for the fp16 overflow counter (below)
                lval = dp.mean_loss_lagged() if lagged else dp.mean_loss()
                if lval is not None:                                 # None = the lagged read has no value yet; a NaN loss is logged as nan (train.py:125-129)
                    n_loss += 1
                    avg_loss = beta * avg_loss + (1 - beta) * lval
                    smoothed_loss = avg_loss / (1 - beta ** (status_every * n_loss))
                if f16:
                    # loss-scale policy of Apex's dynamic scaler at this loop's reporting point: halve on overflow (the kernel
                    # already skipped those steps), double after 2000 clean steps.  Gated on the arithmetic, not on the current
                    # scale: a scale that overflows halved its way down to 1 must be able to grow back.
                    # One process: the counter comes with the lagged scalars (the value of the PREVIOUS reporting point, cumulative, never reset
                    # here), so the policy reacts one interval later and the queue is never drained -- reading it directly stalled the host
                    # every status_every steps.  The interval right after a halving ran partly under the old scale: its overflows are not
                    # counted a second time.  Data parallel: the direct read, beside the collective mean_loss() already is.
                    if lagged:
                        sc = getattr(dp, "lagged_scalars", None)
                        over = 0
                        if sc is not None:
                            cum = int(sc[5]); over = cum - seen_overflows; seen_overflows = cum
                            if skip_overflow_report:
                                over, skip_overflow_report = 0, False
                        have = sc is not None
                    else:
                        over, have = engine.overflow_steps(), True
                    if over:
                        engine.loss_scale = max(engine.loss_scale / 2.0, 1.0); clean_steps = 0
                        skip_overflow_report = lagged
                    elif have:
                        clean_steps += status_every
                        if clean_steps >= 2000:
                            engine.loss_scale = min(engine.loss_scale * 2.0, 2.0 ** 24); clean_steps = 0
                if is_main:
                    print(f"\repoch {epoch+1}/{epochs}, time: {time.time()-first_time:.2f}: lr={lr:.2e},mom={mom:.3f}, "
                          f"data_point {data_point}: loss: {smoothed_loss:.3e}   ", end="")
            lr_in_optimizer = lr                                     # train.py:150: takes effect on the next step
            engine.lr = float(lr)                                    # what optimizer.state_dict() reports, as in the reference
            iter_count += 1
            windows += batch_size
        torch.cuda.synchronize()
        t_last = time.time() - t_ep
        t_train += t_last
        vl_avg = eval_status_save(model, engine, effect, epoch, epochs, lr_in_optimizer, 0.0, device, dataloader_val, logfilename,
                                  first_time, beta, vl_avg, out_checkpointname, False, opt, data_point, smoothed_loss, y_size, sr,
                                  status_every, is_main=is_main)
    if is_main:
        world = dist.get_world_size() if dist.is_initialized() else 1
        per_epoch = windows // max(epochs - int(start_epoch), 1)
        print(f"\nTotal elapsed time for training loop = {time.time() - first_time:.2f}  "
              f"({windows * world / max(t_train, 1e-9):.0f} train windows/s incl. the data feed; last epoch {per_epoch * world / max(t_last, 1e-9):.0f})")
    return None


def train(effect=None, epochs=100, n_data_points=200000, batch_size=20, device=None, plot_every=10, cp_every=25, sr=44100,
          datapath=None, scale_factor=1, shrink_factor=4, apex_opt="O0", target_type="stream", lr_max=1e-4,
          in_checkpointname='modelcheckpoint.tar', compand=False, num_workers=10, device_feed=True, compute_dtype=None,
          resume_optimizer=False):
    """train.py:167-278.  datapath: directory with Train/ and Val/ wav pairs (datasets.AudioFileDataSet, the reference's
    file feed, e.g. the LA2A set of BASELINE configs[3]; pass effect=audio.FileEffect(datapath)); compand: mu-law compand that dataset's audio (datasets.py:218-220).
    apex_opt: "O0" = fp32 (the parity path); "O1" / "O2" / "O3" = the reference's Apex mixed precision (train.py:254-255),
    here float16 operands with fp32 accumulation, a loss scale and the L1 clip over all parameters (train.py:133-136) --
    compute_dtype "f16_all".  Extra keywords (not in the reference): compute_dtype overrides the arithmetic ("f32", "bf16",
    "bf16_all", "f16", "f16_all"; bf16 is the MI355X-native choice and needs no loss scale); device_feed: True (default) = the synthetic task's minibatches are
    generated on the GPU (csrc/st_feed.h; "recycle": one device-resident training set re-sampled each epoch) and file datasets gather their windows on the
    device; False = the reference's feed, a torch DataLoader with `num_workers` CPU workers over the Dataset's __getitem__ (two to three orders of magnitude
    below the step rate; printed when chosen); resume_optimizer: False (default, the reference's behaviour: train `epochs`
    more epochs from the loaded weights with a fresh optimizer and schedule -- its fine-tune workflow); True restores Adam's moments
    from the checkpoint (which the reference saves but never reads back, train.py:229) and, if the checkpoint belongs to THIS schedule
    (its epoch counter is below `epochs` and its step count lies inside the 1-cycle table), also the position in the run."""
    if compute_dtype is None:
        compute_dtype = "f32" if str(apex_opt).upper() in ("O0", "NONE", "") else "f16_all"
    effect = audio.Compressor_4c() if effect is None else effect
    device = torch.device("cuda:0") if device is None else torch.device(device)
    print(f'SignalTrain (MI355X) training execution began at {time.ctime()}. Options:')
    print(f'    epochs = {epochs}, n_data_points = {n_data_points}, batch_size = {batch_size}')
    print(f'    scale_factor = {scale_factor}, shrink_factor = {shrink_factor}')
    num_knobs = len(effect.knob_names)
    effect.info()
    state_dict, rv = misc.load_checkpoint(in_checkpointname, fatal=False, device="cpu")
    if state_dict != {}:
        scale_factor, shrink_factor, sr = rv['scale_factor'], rv['shrink_factor'], rv['sr']
    model = nn_proc.st_model(scale_factor=scale_factor, shrink_factor=shrink_factor, num_knobs=num_knobs, sr=sr)
    if state_dict != {}:
        model.load_state_dict(state_dict)
    chunk_size, out_chunk_size = model.in_chunk_size, model.out_chunk_size
    if n_data_points < batch_size or epochs < 1:
        # the reference fails late and obscurely here (an epoch needs >= 10 minibatches: UnboundLocalError at train.py:158; the 1-cycle table is empty)
        raise ValueError(f"train(): n_data_points = {n_data_points} with batch_size = {batch_size}, epochs = {epochs}: less than one minibatch per epoch -- nothing to train on")
    print("Model defined.  Number of trainable parameters:", sum(p.numel() for p in model.parameters() if p.requires_grad))
    model.to(device)
    model.set_compute_dtype(compute_dtype)
    engine = model.engine(torch.zeros(batch_size, chunk_size, device=device))     # parameters become views of the engine's flat buffer
    seed_data_streams(dist.get_rank() if dist.is_initialized() else 0)            # identical weights above, distinct minibatches below
    start_epoch, start_iter, lr_resume = 0, 0, None
    lr_sched, mom_sched = learningrate.get_1cycle_schedule(lr_max=lr_max, n_data_points=n_data_points, epochs=epochs, batch_size=batch_size)
    if datapath is not None:
        # pre-recorded input / target pairs (train.py:241-246; BASELINE configs[3]): windows gathered on the device from the preloaded audio
        dataset = datasets.AudioFileDataSet(chunk_size, effect, sr=sr, datapoints=n_data_points, path=datapath + "/Train/", y_size=out_chunk_size,
                                            rerun=(target_type != "stream"), augment=True, preload=True, compand=compand)
        dataset_val = datasets.AudioFileDataSet(chunk_size, effect, sr=sr, datapoints=n_data_points // 4, path=datapath + "/Val/", y_size=out_chunk_size,
                                                rerun=(target_type != "stream"), augment=False, compand=compand)
        if device_feed and target_type == "stream":
            class _FileLoader:
                def __init__(self_inner, ds): self_inner.ds = ds
                def __iter__(self_inner): return self_inner.ds.device_batches(batch_size, device)
                def __len__(self_inner): return len(self_inner.ds) // batch_size      # minibatches per epoch (device_batches drops the remainder, like drop_last=True)
            dataloader, dataloader_val = _FileLoader(dataset), _FileLoader(dataset_val)
        else:
            dataloader = DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, shuffle=True, worker_init_fn=datasets.worker_init, drop_last=True)
            dataloader_val = DataLoader(dataset_val, batch_size=batch_size, num_workers=num_workers, shuffle=False, drop_last=True)
    else:
        # synthetic effect: every training minibatch is generated ON the GPU (one st_synth_comp4c launch per minibatch for the comp_4c
        # effects), as the reference's non-recycled dataset does on its CPU workers (train.py:233-248); device_feed="recycle": one dataset
        # generated up front and re-sampled by index each epoch (the reference's recycle=True mode); device_feed=False: the reference's CPU-worker
        # DataLoader (~100 windows/s per core against 3-8 x 10^5 per second for the step).
        # Validation: the reference's recycled set (train.py:237-238), resident in HBM.
        dataset = datasets.SynthAudioDataSet(chunk_size, effect, sr=sr, datapoints=n_data_points, y_size=out_chunk_size, augment=True)
        t0 = time.time()
        if not device_feed:
            # the reference's own feed (train.py:233-248): CPU workers behind a DataLoader, items from SynthAudioDataSet.__getitem__
            print(f"device_feed=False: training minibatches come from {num_workers} CPU DataLoader workers (reference-style feed; expect ~10^2-10^3 windows/s)")
            dataloader = DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, shuffle=True, worker_init_fn=datasets.worker_init, drop_last=True)
        elif device_feed == "recycle":
            dev_ds = datasets.DeviceRecycledDataSet(chunk_size, effect, sr=sr, datapoints=n_data_points, y_size=out_chunk_size, augment=True, device=device)

            class _DevLoader:                      # iterable with the DataLoader's per-epoch semantics
                def __iter__(self_inner):
                    return dev_ds.batches(batch_size, shuffle=True)

                def __len__(self_inner):
                    return n_data_points // batch_size
            dataloader = _DevLoader()
        else:
            dataloader = datasets.DeviceSynthLoader(dataset, batch_size, device)
        val_ds = datasets.DeviceRecycledDataSet(chunk_size, effect, sr=sr, datapoints=n_data_points // 4, y_size=out_chunk_size, augment=False, device=device)

        class _ValLoader:
            def __iter__(self_inner):
                return val_ds.batches(batch_size, shuffle=False)

            def __len__(self_inner):
                return (n_data_points // 4) // batch_size
        dataloader_val = _ValLoader()
        print(f"device-side data feed ready in {time.time() - t0:.1f} s ({n_data_points // 4} validation windows resident in HBM)")
    if state_dict != {} and resume_optimizer and rv.get('optimizer'):
        lr_resume = engine.load_optimizer_state_dict(rv['optimizer'])
        if lr_resume is not None:
            ck_epoch, ck_iter = int(rv.get('epoch', 0)), engine.step_count
            try:
                per_epoch = max(len(dataloader), 1)                 # the loader's own length (file datasets, recycled sets, drop_last)
            except TypeError:
                per_epoch = max(n_data_points // batch_size, 1)
            if ck_epoch < epochs and ck_iter < len(lr_sched) and ck_iter == ck_epoch * per_epoch:
                start_epoch, start_iter = ck_epoch, ck_iter
                print(f"Optimizer state restored: {start_iter} steps done, resuming at epoch {start_epoch + 1} with lr = {lr_resume:.3e}")
            else:
                # a finished run, or a checkpoint of another n_data_points / batch_size / epochs: its position means nothing in this schedule
                print(f"WARNING: the checkpoint's position (epoch {ck_epoch}, step {ck_iter}) does not belong to this schedule ({epochs} epochs of "
                      f"{per_epoch} steps): Adam's moments are kept, epoch / step / learning rate restart at the beginning")
                lr_resume = None
    logfilename = "vl_avg_out.dat"
    open(logfilename, "a").close()
    train_loop(model, engine, effect, device, epochs, batch_size, lr_sched, mom_sched, dataloader, dataloader_val,
               out_chunk_size, logfilename, "modelcheckpoint.tar", sr=sr, lr_max=lr_max,
               start_epoch=start_epoch, start_iter=start_iter, lr_resume=lr_resume)
    return model
