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
