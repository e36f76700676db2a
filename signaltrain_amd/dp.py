"""Data-parallel training: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The reference's only multi-GPU mechanism is a disabled nn.DataParallel stub (train.py:259-263).  Here each
rank runs the fused step on its shard of the minibatch with identical parameters; the flat fp32 gradient
buffer (16.8 MB) is sum-all-reduced in two buckets -- [synthesis bases + both autoencoders] (8.45 MB) as soon as
they are final, overlapping the analysis weight-gradient GEMM, then one contiguous range holding the 513 live rows
of the two analysis tensors (6.3 MB incl. 2 MB of structural zeros) -- scaled by 1/world, and only then L1-clipped (the norm is a function of the reduced gradient, so it
is identical on every rank and needs no second collective) and fed to the replicated Adam.
"""
import torch
import torch.distributed as dist


class DataParallel:
    """Wraps an engine exposing loss_backward_p1/p2, grad_buckets(), clip_adam(), scalars."""

    def __init__(self, engine, process_group=None, force_collectives=False):
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
        eng.loss_backward_p1(x, knobs, y)
        b = eng.grad_buckets()
        # async collective on the communicator's stream: it waits for phase 1 only, phase 2 overlaps it
        h0 = dist.all_reduce(b[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        eng.loss_backward_p2()
        h1 = dist.all_reduce(b[1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h0.wait(); h1.wait()
        return eng.clip_adam(lr, grad_scale=1.0 / self.world, **kw)

    def mean_loss(self):
        """Global-batch loss for logging (the reference reads the loss every 10 iterations, train.py:125)."""
        t = self.engine.scalars[:3].clone()
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t /= self.world
        return float(t[0].item())
