"""Data-parallel step with the REAL engine and world size 2: two processes share the one GPU of the test box and exchange
gradients through gloo (RCCL refuses two ranks on one device; the driver's multi-GPU run uses RCCL with the same code).
Checks the overlapped hand-off (eae_dp_stream) numerically: after every step both ranks hold the SAME parameters, and they equal
what one process gets from the two shards' gradients summed and fed to Adam with grad_scale 1/2 -- bitwise."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, B, ALPHA, LR = 3, 32, 35.0, 5e-3


def _data(rank):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as gu
    x, y = gu.make_images(B, 900 + rank)
    return torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()


def _model():
    sys.path.insert(0, ROOT)
    import eae_amd
    torch.manual_seed(4242)
    return eae_amd.SupervisedAutoencoder(64, 10).cuda()


def _worker(rank, world, initfile, outdir):
    import torch.distributed as dist
    from eae_amd import dp
    from eae_amd.engine import engine_for
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    m = _model(); m.train()
    eng = engine_for(m, max_batch=B)
    tr = dp.DataParallelTrainer(eng)
    tr.broadcast_parameters()
    x, y = _data(rank)
    for s in range(STEPS):
        tr.train_step(x, y, ALPHA, LR)
        torch.cuda.synchronize()
        np.save(os.path.join(outdir, f"p_{rank}_{s}.npy"), eng.params.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_dp_world2_real_engine_on_one_gpu(overlap, monkeypatch):
    import torch.multiprocessing as mp
    from eae_amd.engine import engine_for
    monkeypatch.setenv("EAE_DP_OVERLAP", overlap)
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(2, initfile, d), nprocs=2, join=True)
        got = [[np.load(os.path.join(d, f"p_{r}_{s}.npy")) for s in range(STEPS)] for r in range(2)]
    # single-process reference: per-shard gradients from the same parameters, summed, Adam with grad_scale 1/2
    m = _model(); m.train()
    eng = engine_for(m, max_batch=B)
    shards = [_data(0), _data(1)]
    for s in range(STEPS):
        g = None
        for x, y in shards:
            eng.grad_step(x, y, ALPHA)
            torch.cuda.synchronize()
            g = eng.grads.clone() if g is None else g + eng.grads
        eng.grads.copy_(g)
        eng.adam_step(LR, grad_scale=0.5)
        torch.cuda.synchronize()
        ref = eng.params.cpu().numpy()
        assert np.array_equal(got[0][s], got[1][s]), f"replicas diverged at step {s}"
        assert np.array_equal(got[0][s], ref), f"step {s}: max diff {np.abs(got[0][s] - ref).max()}"


# ---------------------------------------------------------------------------------------------------------------
# SyncBN: 2 ranks x 16 images == 1 rank x 32 images (SURVEY.md 8e "Equivalence to test")
# ---------------------------------------------------------------------------------------------------------------
SB = 32


def _sync_worker(rank, world, initfile, outdir):
    import torch.distributed as dist
    from eae_amd import dp
    from eae_amd.engine import engine_for
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as gu
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    m = _model(); m.train()
    eng = engine_for(m, max_batch=SB)
    tr = dp.DataParallelTrainer(eng, sync_bn=True)
    tr.broadcast_parameters()
    x, y = gu.make_images(SB, 950)
    h = SB // world
    xs, ys = torch.from_numpy(x[rank * h:(rank + 1) * h]).cuda(), torch.from_numpy(y[rank * h:(rank + 1) * h]).cuda()
    z = torch.empty((h, 64), device="cuda")
    eng.grad_step(xs, ys, ALPHA)
    torch.cuda.synchronize()
    assert tr.sync_bn.error is None, tr.sync_bn.error
    loss = eng.loss_last.cpu().numpy().copy()
    dist.all_reduce(eng.grads)
    eng.grads.mul_(1.0 / world)
    torch.cuda.synchronize()
    zz = eng.encoder(xs, train=False)          # eval-mode latents use the running statistics the synchronized step produced
    np.savez(os.path.join(outdir, f"sync_{rank}.npz"), grads=eng.grads.cpu().numpy(), bn=eng.bn_running.cpu().numpy(),
             nbt=eng.bn_nbt.cpu().numpy(), loss=loss, z=zz.cpu().numpy())
    # ... and two optimizer steps through the trainer keep the replicas identical
    for s in range(2):
        tr.train_step(xs, ys, ALPHA, LR)
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"syncp_{rank}.npy"), eng.params.cpu().numpy())
    # uneven shards (a short last batch on one rank) must be refused, on every rank, before any statistics are exchanged
    msg = ""
    try:
        tr.train_step(xs[: h - 4 * rank].contiguous(), ys[: h - 4 * rank].contiguous(), ALPHA, LR)
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(outdir, f"uneven_{rank}.txt"), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_world2_equals_single_process_global_batch():
    import torch.multiprocessing as mp
    import gpu_util as G
    import golden_util as gu
    from eae_amd.engine import engine_for
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_sync_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        r = [np.load(os.path.join(d, f"sync_{k}.npz")) for k in range(2)]
        p = [np.load(os.path.join(d, f"syncp_{k}.npy")) for k in range(2)]
        uneven = [open(os.path.join(d, f"uneven_{k}.txt")).read() for k in range(2)]
    assert all("same per-rank batch size" in u for u in uneven), uneven
    m = _model(); m.train()
    eng = engine_for(m, max_batch=SB)
    x, y = gu.make_images(SB, 950)
    eng.grad_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), ALPHA)
    torch.cuda.synchronize()
    ref_g, ref_bn = eng.grads.cpu().numpy(), eng.bn_running.cpu().numpy()
    ref_loss = eng.loss_last.cpu().numpy()
    # forward statistics are integer sums of the same per-tile partials: the running statistics agree to the last bit
    assert np.array_equal(r[0]["bn"], r[1]["bn"]) and np.array_equal(r[0]["bn"], ref_bn), np.abs(r[0]["bn"] - ref_bn).max()
    assert np.array_equal(r[0]["nbt"], eng.bn_nbt.cpu().numpy())
    # loss: mean of the two shard means == the global mean
    assert abs(0.5 * (r[0]["loss"][0] + r[1]["loss"][0]) - ref_loss[0]) <= 1e-5 * abs(ref_loss[0])
    # gradients: identical on both ranks; equal to the single-process global-batch gradients up to fp32 / fp64 summation order
    # (the backward sums are exchanged in fp64, so a BatchNorm-backward coefficient can differ in its last bit and flip the
    # bf16 rounding of single gradient elements)
    assert np.array_equal(r[0]["grads"], r[1]["grads"])
    off = eng.poff
    for i in range(38):
        a, b = r[0]["grads"][off[i]:off[i + 1]], ref_g[off[i]:off[i + 1]]
        if np.abs(b).max() == 0.0:
            assert np.abs(a).max() == 0.0
            continue
        assert G.relmax(a, b) <= 5e-3 and G.cosine(a, b) > 0.99999, (i, G.relmax(a, b), G.cosine(a, b))
    zz = eng.encoder(torch.from_numpy(x).cuda(), train=False).cpu().numpy()
    assert np.array_equal(np.concatenate([r[0]["z"], r[1]["z"]]), zz)
    assert np.array_equal(p[0], p[1])


def _fit_worker(rank, world, initfile, outdir):
    import json
    import torch.distributed as dist
    from eae_amd import dp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as gu
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    x, y = gu.make_images(96, 960)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)

    def shard_loader(lo, hi, bs):       # global batches of `bs`, this rank's half of each
        out = []
        for i in range(lo, hi, bs):
            j = min(i + bs, hi)
            h = (j - i) // world
            out.append((xt[i + rank * h: i + (rank + 1) * h], yt[i + rank * h: i + (rank + 1) * h]))
        return out

    m = _model()
    r = dp.fit_autoencoder_dp(shard_loader(0, 64, 32), shard_loader(64, 96, 32), ALPHA, LR, model=m, num_epochs=3, patience=15,
                              verbose=False)
    json.dump({"train": r["train_curve"], "val": r["val_curve"], "epochs": r["epochs"]}, open(os.path.join(outdir, f"fit_{rank}.json"), "w"))
    np.save(os.path.join(outdir, f"fitp_{rank}.npy"), m.state_dict()["enc.encoder.0.weight"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_fit_autoencoder_dp_world2_same_curves_on_every_rank():
    """DP-aware fit loop: the epoch means are all-reduced once per epoch phase, so both ranks log the same curves (and would
    stop at the same epoch); the replicas end identical."""
    import json
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_fit_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        a, b = (json.load(open(os.path.join(d, f"fit_{k}.json"))) for k in range(2))
        pa, pb = (np.load(os.path.join(d, f"fitp_{k}.npy")) for k in range(2))
    assert a == b and a["epochs"] == 3 and np.isfinite(a["train"]).all() and np.isfinite(a["val"]).all()
    assert a["train"][-1] < a["train"][0]
    assert np.array_equal(pa, pb)
