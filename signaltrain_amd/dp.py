"""Data-parallel training: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The reference's only multi-GPU mechanism is a disabled nn.DataParallel stub (train.py:259-263).  Here each
rank runs the fused step on its shard of the minibatch with identical parameters; the flat fp32 gradient buffer is
sum-all-reduced range by range, in the order the backward makes the ranges final, each collective running under the
following kernels.  Two schedules (DataParallel(schedule=...)):

  "two_bucket" (default)   phase 1 = forward + loss + synthesis / autoencoder / polar backward
                             -> all-reduce [synthesis bases + autoencoders] (8.45 MB)   || phase 2 = analysis weight gradient
                             -> all-reduce the packed copy of the 513 live rows of both analysis bases (4.2 MB, exposed), copy back
  "staged"                 four stages (engine.loss_backward_stage / stage_bucket): synthesis bases 4.2 MB || autoencoder
                             backward; autoencoders 67 KB; real analysis basis 2.1 MB || imaginary-basis GEMM; imaginary
                             basis 2.1 MB (the only exposed one) -- 10.6 MB instead of 14.7 MB on the wire, at a measured fixed
                             cost of two more collectives and two smaller GEMMs

The reduced gradient is scaled by 1/world and only then L1-clipped (the norm is a function of the reduced gradient, so it is
identical on every rank and needs no second collective) and fed to the replicated Adam.
"""
import torch
import torch.distributed as dist


class DataParallel:
    """Wraps an engine exposing N_STAGES, loss_backward_stage(), stage_bucket(), clip_adam(), scalars."""

    def __init__(self, engine, process_group=None, force_collectives=False, schedule="two_bucket"):
        """schedule: "two_bucket" (default: [synthesis + autoencoders] 8.45 MB under the analysis weight gradient, then the
        packed 4.2 MB of live analysis rows) or "staged" (four stages / four ranges, only the last 2.1 MB exposed, but +55..85 us
        of fixed cost measured on one GPU: two more collectives and the per-basis analysis GEMMs -- see DESIGN.md section 6)."""
        assert schedule in ("staged", "two_bucket"), schedule
        self.schedule = schedule
        self.engine = engine
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: take the bucketed all-reduce path even with one rank (tests exercise the N > 1 code on one GPU)
        self.force = bool(force_collectives) and dist.is_initialized()

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            dist.broadcast(self.engine.params, src=src, group=self.group)

    def train_step(self, x, knobs, y, lr, **kw):
        eng = self.engine
        if self.world == 1 and not self.force:
            return eng.train_step(x, knobs, y, lr, **kw)
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

    def mean_loss(self):
        """Global-batch loss for logging (the reference reads the loss every 10 iterations, train.py:125)."""
        t = self.engine.scalars[:3].clone()
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t /= self.world
        return float(t[0].item())
