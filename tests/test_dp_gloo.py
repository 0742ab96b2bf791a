"""Data-parallel host logic with world_size 2 on CPU (gloo): bucketed gradient all-reduce == mean of per-shard gradients,
replicas stay identical.  The per-rank 'engine' here is a stand-in whose gradients come from the NumPy oracle on that
rank's shard (tests may use the oracle; the product's engine computes them in HIP)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Same attribute surface as eae_amd.engine.AEEngine for what dp.DataParallelTrainer touches."""

    def __init__(self, seed):
        sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
        import eae_amd
        from oracle import ae_numpy as O
        self.O = O
        torch.manual_seed(seed)
        self.model = eae_amd.SupervisedAutoencoder(64)
        self.names = [n for n, _ in self.model.named_parameters()]
        sizes = [p.numel() for p in self.model.parameters()]
        self.poff = [0]
        for n in sizes:
            self.poff.append(self.poff[-1] + ((n + 3) & ~3))
        tot = self.poff[38]
        self.params = torch.zeros(tot); self.grads = torch.zeros(tot)
        self.adam_m = torch.zeros(tot); self.adam_v = torch.zeros(tot)
        self.bn_running = torch.zeros(1408); self.bn_nbt = torch.zeros(7, dtype=torch.int64)
        for i, p in enumerate(self.model.parameters()):
            self.params[self.poff[i]: self.poff[i] + p.numel()] = p.detach().flatten()
        self.t = 0

    def params_changed(self):
        pass

    def _np_params(self):
        sd = {k: v.numpy().copy() for k, v in self.model.state_dict().items()}
        for i, (n, p) in enumerate(self.model.named_parameters()):
            sd[n] = self.params[self.poff[i]: self.poff[i] + p.numel()].view(p.shape).numpy().copy()
        return sd

    def grad_step(self, x, labels, alpha, head=True):
        p = self._np_params()
        out = self.O.ae_forward(p, x.numpy(), True)
        g = self.O.ae_backward(p, out, x.numpy(), labels.numpy(), alpha)
        for i, n in enumerate(self.names):
            self.grads[self.poff[i]: self.poff[i] + g[n].size] = torch.from_numpy(np.ascontiguousarray(g[n]).ravel())

    def adam_step(self, lr, weight_decay=0.0):
        self.t += 1
        b1, b2 = 0.9, 0.999
        self.adam_m.mul_(b1).add_(self.grads, alpha=1 - b1)
        self.adam_v.mul_(b2).addcmul_(self.grads, self.grads, value=1 - b2)
        denom = self.adam_v.sqrt() / (1 - b2 ** self.t) ** 0.5 + 1e-8
        self.params.addcdiv_(self.adam_m, denom, value=-lr / (1 - b1 ** self.t))


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden_util as gu
        from eae_amd import dp
        torch.set_num_threads(2)
        eng = OracleEngine(seed=100 + rank)            # deliberately different initial replicas
        tr = dp.DataParallelTrainer(eng)
        # buckets tile the arena exactly, in backward order (last tensors first)
        spans = sorted(tr.buckets)
        assert spans[0][0] == 0 and spans[-1][1] == eng.poff[38] and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert tr.buckets[0][1] == eng.poff[38]
        tr.broadcast_parameters(src=0)
        ref = eng.params.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, eng.params)
        x, y = gu.make_images(4, 900)                  # global batch of 4 -> shards of 2
        xs, ys = torch.from_numpy(x[rank * 2:(rank + 1) * 2]), torch.from_numpy(y[rank * 2:(rank + 1) * 2])
        eng.grad_step(xs, ys, 35.0)
        local = eng.grads.clone()
        tr.allreduce_gradients()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        expect = (gathered[0] + gathered[1]) / world
        assert torch.allclose(eng.grads, expect, rtol=1e-6, atol=1e-9)
        eng.adam_step(1e-3)
        tr.train_step(xs, ys, 35.0, 1e-3)              # a second, full DP step
        mine = eng.params.clone()
        other = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        assert torch.equal(other[0], other[1])         # replicas stay bit-identical
        if rank == 0:
            open(os.path.join(tmp, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp_world2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok")


class _FakeLib:
    """libeae stand-in for the communicator entry points.  `fail_at`: ("init", r) -- rank r cannot join (eae_dp_init fails like a refused
    communicator would); ("unique_id", r) -- rank r cannot even reach the library (a missing librccl symbol); None -- every rank joins.
    The all-reduce stand-in is a REAL collective (gloo all-reduce of the gradient words): a rank that entered it while a peer never
    does would hang the test, which is exactly what the bring-up protocol has to exclude."""

    def __init__(self, rank, fail_at, eng):
        self.rank, self.fail_at, self.destroyed, self.eng, self.world_now, self.allreduces = rank, fail_at, False, eng, 0, 0

    def eae_dp_world(self, ctx):
        return self.world_now

    def eae_dp_unique_id(self, buf):
        if self.fail_at == ("unique_id", self.rank):
            return -5
        for i in range(128):
            buf[i] = (i * 7 + 1) & 255
        return 0

    def eae_dp_init(self, ctx, rank, world, raw):
        assert bytes(raw) == bytes(((i * 7 + 1) & 255) for i in range(128)), "every rank must receive rank 0's id"
        if self.fail_at == ("init", rank):
            return -5
        self.world_now = world
        return 0

    def eae_dp_allreduce_bucket(self, ctx, stream, off, count):
        self.allreduces += 1
        lo, n = int(off.value), int(count.value)
        dist.all_reduce(self.eng.grads[lo:lo + n], op=dist.ReduceOp.SUM)
        return 0

    def eae_dp_destroy(self, ctx):
        self.destroyed = True
        self.world_now = 0
        return 0

    def eae_last_error(self):
        return b"simulated: the communicator could not be brought up"


def _worker_fallback(rank, world, port, tmp, fail_at):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from eae_amd import dp, _lib
        eng = OracleEngine(seed=7)
        eng.ctx, eng.device = object(), torch.device("cpu")
        eng.lib = _FakeLib(rank, fail_at, eng)
        _lib.load = lambda: eng.lib                          # check() reads the error text through the loaded library
        real_backend = dist.get_backend
        dist.get_backend = lambda group=None: "nccl"         # take the native branch (the collectives underneath stay gloo)
        try:
            tr = dp.DataParallelTrainer(eng, native=True)
        finally:
            dist.get_backend = real_backend
        flags = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(flags, torch.tensor([1.0 if tr.native else 0.0]))
        if fail_at is None:
            # every rank joined and the known-vector all-reduce came out right on every rank: the engine-owned exchange is trusted
            assert tr.native is True and all(float(f) == 1.0 for f in flags) and eng.lib.allreduces == 1 and not eng.lib.destroyed
            assert tr.rccl_ranks() == world
        else:
            # ONE rank failed: EVERY rank has left the native path (a half-native group would hang in its first collective), and no
            # rank has entered the engine's all-reduce -- the good ranks would have waited there for ever
            assert tr.native is False and tr.rccl_ranks() == 0 and eng.lib.destroyed and eng.lib.allreduces == 0
            assert all(float(f) == 0.0 for f in flags)
            # ... and the torch.distributed exchange works from there
            import golden_util as gu
            tr.broadcast_parameters(src=0)
            x, y = gu.make_images(4, 901)
            tr.train_step(torch.from_numpy(x[rank * 2:(rank + 1) * 2]), torch.from_numpy(y[rank * 2:(rank + 1) * 2]), 35.0, 1e-3)
            mine = eng.params.clone()
            other = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(other, mine)
            assert torch.equal(other[0], other[1])
        if rank == 0:
            open(os.path.join(tmp, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("fail_at", [("init", 1), ("init", 0), ("unique_id", 0), ("unique_id", 1), None],
                         ids=["init-fails-on-rank1", "init-fails-on-rank0", "rank0-cannot-draw-an-id", "rank1-cannot-reach-rccl", "all-join"])
def test_native_exchange_failure_on_one_rank_makes_every_rank_fall_back(tmp_path, fail_at):
    """dp.DataParallelTrainer brings the engine-owned RCCL communicator up with the SAME sequence of collectives on every rank whatever
    fails locally, and proves it on a known vector; if ANY rank fails at ANY stage, ALL ranks take the torch.distributed path and none
    has entered the engine's all-reduce (the stand-in's is a real collective: entering it alone hangs the test).  Two gloo ranks."""
    port = 31500 + (os.getpid() % 2000) + 7 * (["init1", "init0", "id0", "id1", "none"].index(
        {("init", 1): "init1", ("init", 0): "init0", ("unique_id", 0): "id0", ("unique_id", 1): "id1", None: "none"}[fail_at]))
    mp.spawn(_worker_fallback, args=(2, port, str(tmp_path), fail_at), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok")


def test_bucket_bounds_cover_arena():
    from eae_amd import dp
    offs = list(range(0, 39 * 8, 8))
    for nb in (1, 2, 3):
        b = dp.bucket_bounds(offs, offs[38], nb)
        assert len(b) == nb
        spans = sorted(b)
        assert spans[0][0] == 0 and spans[-1][1] == offs[38]
        assert all(x[1] == y[0] for x, y in zip(spans, spans[1:]))
