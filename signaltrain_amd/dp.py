"""Data-parallel training: one process per GPU, RCCL over xGMI.

The reference's only multi-GPU mechanism is a disabled nn.DataParallel stub (train.py:259-263).  Here each rank runs the
step on its shard of the minibatch with identical parameters; the flat fp32 gradient is sum-all-reduced bucket by bucket,
in the order the backward makes the buckets final, each collective running under the kernels that follow it; the reduced
gradient is scaled by 1/world and only then L1-clipped (the norm is a function of the reduced gradient, so it is identical on
every rank and needs no second collective) and fed to the replicated Adam.

Two back ends:

  backend="lib" (default on GPUs)   the exchange lives in libsignaltrain_hip.so (st_dp_*): the library owns the RCCL
        communicator and a communicator stream, and ONE C call (st_dp_train_step) runs phase 1 -> all-reduce [synthesis bases
        + autoencoders] (8.45 MB) || phase 2 (analysis weight gradient) -> all-reduce the packed live analysis rows (4.2 MB,
        the only exposed one) -> clip + Adam.  No Python between the buckets.  torch.distributed (any backend, gloo is enough) is
        used only as the bootstrap channel for the 128-byte RCCL unique id.  schedule="staged" splits that last exchange by basis (the real
        rows' all-reduce under the GEMM of the imaginary rows: 2.1 MB exposed); pack16=True moves it as bfloat16 in the *_all arithmetic modes -- OPT-IN and lossy beyond the mode's own rounding: RCCL reduces in the payload type, so a ring over W ranks
        rounds the partial sum W - 1 times (relative error of the analysis-basis gradient and of the clip norm derived from it ~ 2^-9 sqrt(W - 1): ~1e-2 at W = 8, 2-3
        significant digits).  The two-rank tests model that per-hop rounding (tests/fake_rccl.cpp); check convergence on the real node before relying on it.
  backend="torch"                   the same protocol driven from Python over torch.distributed collectives (backend "nccl" ==
        RCCL, or "gloo" on CPU for the tests), with two schedules: "two_bucket" (as above) and "staged" (four stages / four
        ranges, only the last 2.1 MB exposed; measured +55..85 us of fixed cost on one GPU).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


def library_communicator(device, group=None):
    """Create the library-owned RCCL communicator for this process (collective over all ranks of `group`): rank 0 makes the
    unique id (st_dp_unique_id), torch.distributed carries it to the others, every rank calls st_dp_init.  Returns st_dp*."""
    lib = _lib.load()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    buf = (C.c_char * 128)()
    if rank == 0:
        _lib.check(lib.st_dp_unique_id(buf), "st_dp_unique_id")
    if world > 1:
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0, group=group)
        buf.raw = box[0]
    handle = C.c_void_p()
    with torch.cuda.device(device):
        _lib.check(lib.st_dp_init(buf, int(rank), int(world), C.byref(handle)), "st_dp_init")
    return handle


class DataParallel:
    """Wraps an engine exposing train_step / loss_backward_p1 / _p2 / grad_buckets / finish_buckets / clip_adam
    (+ loss_backward_stage / stage_bucket for the staged schedule, dp_train_step for the library back end)."""

    def __init__(self, engine, process_group=None, force_collectives=False, schedule="two_bucket", backend=None, pack16=False):
        assert schedule in ("staged", "two_bucket"), schedule
        self.schedule = schedule
        self.pack16 = bool(pack16)
        self.engine = engine
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: take the exchange path even with one rank (tests exercise the N > 1 code on one GPU)
        self.force = bool(force_collectives)
        on_gpu = hasattr(engine, "dp_train_step") and getattr(getattr(engine, "device", None), "type", "cpu") == "cuda"
        if backend is None:
            backend = "lib" if on_gpu else "torch"
        assert backend in ("lib", "torch"), backend
        if backend == "lib" and not on_gpu:
            raise RuntimeError("DataParallel(backend='lib') needs the HIP engine on a ROCm device")
        if backend == "torch" and self.force and not dist.is_initialized():
            raise RuntimeError("DataParallel(backend='torch', force_collectives=True) needs an initialised process group")
        self.backend = backend
        if backend == "lib" and (self.world > 1 or self.force) and engine.dp is None:
            engine.dp = library_communicator(engine.device, process_group)

    def close(self):
        eng = self.engine
        if self.backend == "lib" and getattr(eng, "dp", None) is not None:
            _lib.check(_lib.load().st_dp_destroy(eng.dp), "st_dp_destroy"); eng.dp = None

    def broadcast_parameters(self, src=0):
        if self.world == 1:
            return
        eng = self.engine
        if self.backend == "lib":
            lib = _lib.load()
            with torch.cuda.device(eng.device):
                st = eng._stream()
                _lib.check(lib.st_dp_broadcast(eng.dp, _lib.ptr(eng.params), eng.params.numel(), int(src), st), "st_dp_broadcast")
                _lib.check(lib.st_dp_sync(eng.dp, st), "st_dp_sync")
        else:
            dist.broadcast(eng.params, src=src, group=self.group)

    def train_step(self, x, knobs, y, lr, **kw):
        eng = self.engine
        if self.world == 1 and not self.force:
            return eng.train_step(x, knobs, y, lr, **kw)
        if self.backend == "lib":
            return eng.dp_train_step(x, knobs, y, lr, force_exchange=self.force, split_last=(self.schedule == "staged"), pack16=self.pack16, **kw)
        if self.schedule == "two_bucket":
            eng.loss_backward_p1(x, knobs, y)
            b = eng.grad_buckets()
            h0 = dist.all_reduce(b[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            eng.loss_backward_p2()
            h1 = dist.all_reduce(b[1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            h0.wait(); h1.wait()
            eng.finish_buckets()
            return eng.clip_adam(lr, grad_scale=1.0 / self.world, **kw)
        handles = []
        for s in range(eng.N_STAGES):
            eng.loss_backward_stage(s, x, knobs, y)
            # async collective on the communicator's stream: it waits for the stages issued so far, the next stage overlaps it
            handles.append(dist.all_reduce(eng.stage_bucket(s), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        return eng.clip_adam(lr, grad_scale=1.0 / self.world, **kw)

    def mean_loss_lagged(self):
        """The loss for the progress line WITHOUT draining the GPU queue: the scalars of this call are copied to pinned host memory behind the
        work issued so far, and the value RETURNED is the one requested by the previous call (i.e. it lags by one reporting interval, 10
        steps in train_loop; None on the first call -- "no value yet", so that a genuinely NaN loss still reaches the log).  The reference reads loss.item() every 10 batches (train.py:125), which stalls the
        host until the queue is empty -- at 0.3 ms per step that stall is a tenth of the loop.  Single-process only (data parallel needs
        the collective of mean_loss)."""
        eng = self.engine
        if self.world > 1:
            return self.mean_loss()
        if getattr(self, "_pin", None) is None:
            self._pin = [torch.zeros(8, dtype=torch.float32).pin_memory(), torch.zeros(8, dtype=torch.float32).pin_memory()]
            self._pin_ev = [None, None]; self._pin_i = 0
        i = self._pin_i
        prev = None
        j = 1 - i
        self.lagged_scalars = None
        if self._pin_ev[j] is not None:
            self._pin_ev[j].synchronize()                # recorded >= one interval ago: long done
            prev = float(self._pin[j][0])
            self.lagged_scalars = self._pin[j].clone()   # all eight scalars of that reporting point ([5] = steps skipped on overflow so far): the fp16 loss-scale policy reads them here
        with torch.cuda.device(eng.device):
            self._pin[i].copy_(eng.scalars, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(eng.device))
        self._pin_ev[i] = ev; self._pin_i = j
        return prev

    def mean_loss(self):
        """Global-batch loss for logging (the reference reads the loss every 10 iterations, train.py:125)."""
        t = self.engine.scalars[:3].clone()
        if self.world > 1:
            if self.backend == "lib":
                eng, lib = self.engine, _lib.load()
                t = torch.cat([t, t.new_zeros(1)])
                with torch.cuda.device(eng.device):
                    st = eng._stream()
                    _lib.check(lib.st_dp_allreduce(eng.dp, _lib.ptr(t), 4, st), "st_dp_allreduce")
                    _lib.check(lib.st_dp_sync(eng.dp, st), "st_dp_sync")
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t = t / self.world
        return float(t[0].item())
