"""CPU, world_size 2, gloo: the data-parallel step (dp.DataParallel) -- shard, bucketed sum all-reduce of the flat
gradient, 1/world scaling, clip AFTER the reduction, replicated Adam -- gives the same parameters as one process
on the global batch.  The per-rank 'engine' here is the oracle (no GPU in this container) behind the SAME protocol as
signaltrain_amd.engine.StepEngine: phase 1 publishes only the ranges that are final after it (synthesis bases + autoencoders),
phase 2 the analysis weight gradient -- into grads AND packed into the staging buffer the second all-reduce moves -- and
finish_buckets copies the reduced rows back; a range that is all-reduced before its phase has run, or a missing copy-back,
changes the result.  The HIP engine runs the same protocol on the GPU (test_dp_collective_path_single_rank) and from C
(st_dp_train_step)."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import st_oracle as O
from tests.golden_util import perturb_stft


class OracleEngine:
    """Same protocol as signaltrain_amd.engine.StepEngine, arithmetic by the numpy oracle."""

    def __init__(self, P, geo):
        self.geo = geo
        self.keys = O.param_order()
        self.sizes = [P[k].size for k in self.keys]
        self.params = torch.from_numpy(np.concatenate([P[k].ravel() for k in self.keys]).astype(np.float32))
        self.grads = torch.zeros_like(self.params); self.m = torch.zeros_like(self.params); self.v = torch.zeros_like(self.params)
        self.scalars = torch.zeros(8); self.step_count = 0

    def _dict(self, flat):
        out, o = {}, 0
        for k, n in zip(self.keys, self.sizes):
            shp = (1024, 1, 1024) if n == 1 << 20 else None
            out[k] = flat[o:o + n].numpy().reshape(shp) if shp else flat[o:o + n].numpy(); o += n
        return out

    N2, LIVE = 1 << 20, 513 * 1024

    def _backward(self, x, knobs, y):
        P = self._dict(self.params)
        shapes = {k: v.shape for k, v in O.init_params(self.geo, 4).items()}
        P = {k: P[k].reshape(shapes[k]) for k in P}
        loss, G, _ = O.model_loss_bwd(x.numpy(), knobs.numpy(), y.numpy(), P, self.geo)
        self.scalars[0] = float(loss)
        return torch.from_numpy(np.concatenate([G[k].ravel() for k in self.keys]).astype(np.float32))

    def loss_backward_p1(self, x, knobs, y):
        """Forward + backward up to (excluding) the analysis weight gradient: only grads[2 N^2:] is final afterwards; the
        analysis range holds garbage until phase 2 (as on the device, where it still holds the previous step's values)."""
        full = self._backward(x, knobs, y)
        self._analysis = full[:2 * self.N2].clone()
        self.grads[2 * self.N2:] = full[2 * self.N2:]
        self.grads[:2 * self.N2] = float("nan")

    def loss_backward_p2(self):
        """Analysis weight gradient: rows [0,F) of both bases into grads and packed into the staging buffer [2F][N]."""
        n, live = self.N2, self.LIVE
        self.grads[:2 * n] = self._analysis
        if getattr(self, "stage", None) is None:
            self.stage = torch.zeros(2 * live)
        self.stage[:live] = self.grads[0:live]; self.stage[live:] = self.grads[n:n + live]

    N_STAGES = 4

    def loss_backward_stage(self, stage, x=None, knobs=None, y=None):
        """Four stages, each leaving stage_bucket(stage) final (the others are poisoned until their stage has run)."""
        n, live = self.N2, self.LIVE
        if stage == 0:
            self._full = self._backward(x, knobs, y)
            self.grads[:] = float("nan")
        lo, hi = ((2 * n, 4 * n), (4 * n, self.grads.numel()), (0, n), (n, 2 * n))[stage]
        self.grads[lo:hi] = self._full[lo:hi]

    def stage_bucket(self, stage):
        n, live, e = self.N2, self.LIVE, self.grads.numel()
        return (self.grads[2 * n:4 * n], self.grads[4 * n:e], self.grads[0:live], self.grads[n:n + live])[stage]   # as StepEngine.stage_bucket

    def grad_buckets(self):
        if getattr(self, "stage", None) is None:
            self.stage = torch.zeros(2 * self.LIVE)
        return [self.grads[2 * self.N2:], self.stage]                       # as StepEngine.grad_buckets(): [synthesis + autoencoders], packed live analysis rows

    def finish_buckets(self):
        n, live = self.N2, self.LIVE
        self.grads[0:live] = self.stage[:live]; self.grads[n:n + live] = self.stage[live:]

    def clip_adam(self, lr, grad_scale=1.0, **kw):
        self.step_count += 1
        shapes = {k: v.shape for k, v in O.init_params(self.geo, 4).items()}
        G = {k: (v * np.float32(grad_scale)).reshape(shapes[k]) for k, v in self._dict(self.grads).items()}
        O.clip_l1_stft(G)
        P = {k: v.reshape(shapes[k]) for k, v in self._dict(self.params).items()}
        M = {k: v.reshape(shapes[k]) for k, v in self._dict(self.m).items()}
        V = {k: v.reshape(shapes[k]) for k, v in self._dict(self.v).items()}
        O.adam_step(P, G, M, V, self.step_count, lr)
        for flat, D in ((self.params, P), (self.m, M), (self.v, V)):
            flat.copy_(torch.from_numpy(np.concatenate([D[k].ravel() for k in self.keys])))

    def train_step(self, x, knobs, y, lr, **kw):
        self.loss_backward_p1(x, knobs, y); self.loss_backward_p2()
        return self.clip_adam(lr)


def _case():
    geo = O.geometry(1, 4)
    rng = np.random.default_rng(11)
    P = O.init_params(geo, 4, rng); perturb_stft(P, seed=2)
    X, Y, KN = O.synth_comp4c_batch(4, geo["L"], geo["y"], rng)
    return geo, P, X, Y, KN


def _worker(rank, world, port, q, schedule="two_bucket"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from signaltrain_amd.dp import DataParallel
    geo, P, X, Y, KN = _case()
    eng = OracleEngine(P, geo)
    dp = DataParallel(eng, schedule=schedule)
    dp.broadcast_parameters()
    sh = slice(rank * 2, rank * 2 + 2)                      # each rank takes its shard of the global batch
    for it in range(2):
        dp.train_step(torch.from_numpy(X[sh]), torch.from_numpy(KN[sh]), torch.from_numpy(Y[sh]), 1e-3)
    loss = dp.mean_loss()
    if rank == 0:
        q.put((eng.params.numpy().copy(), loss))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("schedule", ["staged", "two_bucket"])
def test_dp_world2_equals_single_process_on_global_batch(schedule):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if schedule == "two_bucket" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, schedule)) for r in range(2)]
    for p in procs:
        p.start()
    got, loss2 = q.get(timeout=240)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    geo, P, X, Y, KN = _case()
    ref = OracleEngine(P, geo)
    for it in range(2):
        ref.train_step(torch.from_numpy(X), torch.from_numpy(KN), torch.from_numpy(Y), 1e-3)
    # fp32 reassociation tolerance: <= 1e-5 on the parameters after the steps
    assert np.abs(got - ref.params.numpy()).max() < 1e-5
    # mean_loss() = the global-batch loss of the LAST step = the single-process loss on the global batch (both terms are means over equal shards)
    assert abs(loss2 - float(ref.scalars[0])) < 1e-4 * abs(float(ref.scalars[0])) + 1e-7, (loss2, float(ref.scalars[0]))
