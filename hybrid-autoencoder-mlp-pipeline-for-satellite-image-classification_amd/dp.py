"""Data-parallel training: one process per GPU, gradients averaged with RCCL over xGMI (torch.distributed 'nccl').

The reference has no parallelism of any kind; this is new work required by BASELINE.json's north_star.
Every rank holds a full replica (1.3 M parameters) and a shard of the global minibatch; per step the flat fp32 gradient
arena (5.26 MB) is all-reduced in a few large buckets and divided by the world size, then the fused Adam runs on every
rank (identical updates keep the replicas in sync).  BatchNorm uses per-rank batch statistics (DDP semantics).

The collective goes through ``torch.distributed`` so that the same code runs on RCCL (GPU) and gloo (CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def bucket_bounds(offsets, total, n_buckets=3):
    """Split the flat arena [0,total) into `n_buckets` contiguous buckets on tensor boundaries, in BACKWARD order of
    completion: classifier + decoder first, then the two latent projections, then the encoder
    (backward runs classifier -> deconv4..1 -> dec.fc -> enc.fc -> conv4..1)."""
    # arena order is forward order: enc convs [0,16), enc.fc 16-17, dec.fc 18-19, deconvs 20-33, classifier 34-37
    cuts = [offsets[16], offsets[20]]
    if n_buckets <= 1:
        return [(0, total)]
    if n_buckets == 2:
        return [(offsets[16], total), (0, offsets[16])]
    return [(cuts[1], total), (cuts[0], cuts[1]), (0, cuts[0])]


class DataParallelTrainer:
    def __init__(self, engine, process_group=None, n_buckets=3):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch one process per GPU with torch.distributed.run)")
        self.eng = engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.buckets = bucket_bounds(engine.poff, engine.poff[38], n_buckets)

    def broadcast_parameters(self, src=0):
        """Identical initial replicas: parameters, BatchNorm running stats and Adam state from rank `src`."""
        for t in (self.eng.params, self.eng.bn_running, self.eng.bn_nbt, self.eng.adam_m, self.eng.adam_v):
            dist.broadcast(t, src=src, group=self.pg)
        self.eng.params_changed()

    def allreduce_gradients(self):
        handles = []
        for lo, hi in self.buckets:
            handles.append(dist.all_reduce(self.eng.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        for h in handles:
            h.wait()
        self.eng.grads.mul_(1.0 / self.world)

    def train_step(self, x, labels, alpha, lr, head=True):
        self.eng.grad_step(x, labels, alpha, head=head)
        self.allreduce_gradients()
        self.eng.adam_step(lr)
