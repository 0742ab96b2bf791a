"""Data-parallel training: one process per GPU, gradients averaged with RCCL over xGMI (torch.distributed 'nccl').

The reference has no parallelism of any kind; this is new work required by BASELINE.json's north_star.
Every rank holds a full replica (1.3 M parameters) and a shard of the global minibatch; per step the flat fp32 gradient
arena (5.26 MB) is all-reduced in a few large buckets and divided by the world size, then the fused Adam runs on every
rank (identical updates keep the replicas in sync).  BatchNorm uses per-rank batch statistics (DDP semantics).

The collective goes through ``torch.distributed`` so that the same code runs on RCCL (GPU) and gloo (CPU tests).

Default exchange: ONE all-reduce of the whole arena after the backward (``EAE_DP_OVERLAP=0`` semantics).  ``EAE_DP_OVERLAP=1``
hands the decoder-side bucket to RCCL while the encoder half of the backward is still running; it stays
opt-in until it has run on more than one rank: the repo's co-residency finding (wrong low lanes of ``v_pk_*_f32 op_sel:``
forms beside MFMA kernels) rests on black-box A/B runs, and although RCCL's gfx950 code contains no such instruction
(``build.scan_packed_fp32_rccl``: 325 packed-FP32 instructions, 0 with ``op_sel:``) that is evidence, not validation.

``sync_bn=True`` sums the BatchNorm batch statistics over the replicas (forward: the fixed-point accumulators, exact; backward:
the fp64 sums), so that R ranks x B/R images reproduce one rank x B images; the default keeps per-rank statistics.
"""
from __future__ import annotations

import os
import sys

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


class SyncBatchNorm:
    """Hooks the engine's BatchNorm statistics into an all-reduce over `process_group` (eae_set_sync_bn, include/eae.h)."""

    def __init__(self, engine, process_group=None):
        import ctypes as C
        from ._lib import SYNC_FN, check
        self.eng, self.pg = engine, process_group
        self.world = dist.get_world_size(process_group)
        n = int(engine.lib.eae_sync_bn_acc_elems(engine.ctx))
        self.acc = torch.zeros(n, dtype=torch.int64, device=engine.device)
        self.sums = torch.zeros(7 * 2 * 256, dtype=torch.float64, device=engine.device)
        self.error = None
        self._cb = SYNC_FN(self._exchange)       # keep the ctypes thunk alive as long as the engine may call it
        with torch.cuda.device(engine.device):
            check(engine.lib.eae_set_sync_bn(engine.ctx, self.world, C.cast(self._cb, C.c_void_p), None,
                                             C.c_void_p(self.acc.data_ptr()), C.c_void_p(self.sums.data_ptr())))
        engine._sync_bn = self

    def _exchange(self, user, kind, off, count, stream):
        try:
            t = (self.acc if kind == 0 else self.sums)[off: off + count]
            dev = self.eng.device
            s = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
            with torch.cuda.stream(s):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            return 0
        except Exception as e:      # never let an exception cross the C boundary: the engine reports EAE_ERR_STATE, we re-raise
            self.error = e
            return -1

    def close(self):
        from ._lib import check
        with torch.cuda.device(self.eng.device):
            check(self.eng.lib.eae_set_sync_bn(self.eng.ctx, 1, None, None, None, None))
        self.eng.params_changed()


class DataParallelTrainer:
    """native=True (the default on the nccl backend with the HIP engine; EAE_DP_NATIVE=0 switches it off): the gradient exchange
    belongs to the ENGINE -- eae_dp_init joins an RCCL communicator of its own (the id travels over torch.distributed once) and
    eae_ae_dp_train_step enqueues forward, backward, the bucketed all-reduce and Adam in one call, the decoder-side bucket overlapping
    the encoder half of the backward (EAE_DP_OVERLAP=0: one all-reduce after the backward).  Otherwise -- gloo (the CPU tests), engines
    without a C context -- the exchange goes through torch.distributed as before."""

    def __init__(self, engine, process_group=None, n_buckets=3, sync_bn=False, native=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch one process per GPU with torch.distributed.run)")
        self.eng = engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.buckets = bucket_bounds(engine.poff, engine.poff[38], n_buckets)
        self._comm = None                      # high-priority stream the collectives are enqueued from (GPU engines only)
        self.sync_bn = SyncBatchNorm(engine, process_group) if (sync_bn and self.world > 1 and hasattr(engine, "ctx")) else None
        self.native = False
        if native is None:
            native = os.environ.get("EAE_DP_NATIVE", "1") == "1"
        if native and hasattr(engine, "ctx") and dist.get_backend(process_group) == "nccl":
            self._init_native_checked()

    def _init_native_checked(self):
        """Join the engine's communicator, then prove it on a known vector (rank r contributes r + 1 to 64 gradient words) before it is
        trusted with real gradients.  Every rank learns whether EVERY rank succeeded (one torch all-reduce of a flag): on any failure
        all ranks fall back to the torch.distributed exchange together -- a half-native group would hang in its first collective."""
        import ctypes as C
        from .engine import _stream
        eng = self.eng
        ok, why = 1, ""
        try:
            self._init_native()
            keep = eng.grads[:64].clone()
            eng.grads[:64] = float(self.rank + 1)
            with torch.cuda.device(eng.device):
                rc = int(eng.lib.eae_dp_allreduce_bucket(eng.ctx, _stream(), C.c_longlong(0), C.c_longlong(64)))
            torch.cuda.synchronize(eng.device)
            want = self.world * (self.world + 1) / 2.0
            if rc != 0 or not bool((eng.grads[:64] == want).all()):
                ok, why = 0, f"self-check of the engine's all-reduce failed (rc {rc}, got {float(eng.grads[0])}, want {want})"
            eng.grads[:64] = keep
        except Exception as e:      # a missing librccl symbol, a refused communicator, ...
            ok, why = 0, f"{type(e).__name__}: {e}"
        flag = torch.tensor([ok], dtype=torch.int32, device=eng.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
        if int(flag.item()) != 1:
            if why:
                print(f"[eae dp] rank {self.rank}: engine-owned RCCL exchange unavailable ({why}); using torch.distributed", file=sys.stderr, flush=True)
            try:
                eng.lib.eae_dp_destroy(eng.ctx)
            except Exception:
                pass
            self.native = False

    def _init_native(self):
        import ctypes as C
        from ._lib import check
        eng = self.eng
        if int(eng.lib.eae_dp_world(eng.ctx)) == self.world:       # the engine already belongs to a communicator of this size
            self.native = True
            return
        idbuf = (C.c_ubyte * 128)()
        t = torch.zeros(128, dtype=torch.uint8, device=eng.device)
        if self.rank == 0:
            check(eng.lib.eae_dp_unique_id(idbuf))
            t.copy_(torch.tensor(list(idbuf), dtype=torch.uint8))
        src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
        dist.broadcast(t, src=src, group=self.pg)
        raw = bytes(t.cpu().tolist())
        with torch.cuda.device(eng.device):
            check(eng.lib.eae_dp_init(eng.ctx, self.rank, self.world, raw))
        self.native = True

    def rccl_ranks(self):
        """Ranks of the engine-owned RCCL communicator (0: the exchange goes through torch.distributed)."""
        return int(self.eng.lib.eae_dp_world(self.eng.ctx)) if self.native else 0

    def broadcast_parameters(self, src=0):
        """Identical initial replicas: parameters, BatchNorm running stats and Adam state from rank `src`."""
        for t in (self.eng.params, self.eng.bn_running, self.eng.bn_nbt, self.eng.adam_m, self.eng.adam_v):
            dist.broadcast(t, src=src, group=self.pg)
        self.eng.params_changed()

    def allreduce_gradients(self):
        """Sum the gradient arena over the ranks (buckets in backward order) and scale by 1/world."""
        handles = []
        for lo, hi in self.buckets:
            handles.append(dist.all_reduce(self.eng.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        for h in handles:
            h.wait()
        self.eng.grads.mul_(1.0 / self.world)

    def train_step(self, x, labels, alpha, lr, head=True):
        """One data-parallel iteration.  With the HIP engine the all-reduce of the decoder-side gradients (tensors 18..37, the
        first bucket in backward order) is enqueued behind the engine's side stream as soon as that half of the backward has
        been launched, so RCCL moves it over xGMI while the encoder half is still computing; the encoder-side bucket follows
        after the second half.  The 1/world scaling is folded into the Adam kernel."""
        eng = self.eng
        if self.sync_bn is not None:
            # SyncBN multiplies the per-rank element count by the world size (eae_api.hip fold_consumer / bn_bwd_fin): every rank must
            # feed the same number of images, or mean / variance / the backward terms are silently wrong.  One tiny
            # collective more in a mode that already issues 14 per step.
            b = int(x.shape[0])
            t = torch.tensor([b, -b], dtype=torch.int64, device=x.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
            hi, lo = int(t[0]), -int(t[1])
            if hi != lo:
                raise RuntimeError(f"SyncBatchNorm needs the same per-rank batch size on every rank (this step: {lo}..{hi}); pad or drop "
                                   "the short last batch (sampler with drop_last / padding), or train without sync_bn")
        if self.native:
            import ctypes as C
            from ._lib import check
            from .engine import _stream
            io, keep = eng._io(x, labels, True, head, alpha)
            overlap = 0 if os.environ.get("EAE_DP_OVERLAP", "1") == "0" else 1
            with torch.cuda.device(eng.device):
                check(eng.lib.eae_ae_dp_train_step(eng.ctx, _stream(), C.byref(io), float(lr), overlap))
            return
        side = eng.side_stream() if hasattr(eng, "grad_step_begin") and hasattr(eng, "side_stream") else None
        if side is None:                       # engines without the split API (CPU stand-in of the gloo test)
            eng.grad_step(x, labels, alpha, head=head)
            self.allreduce_gradients()
            eng.adam_step(lr)
            return
        if os.environ.get("EAE_DP_OVERLAP", "0") != "1":     # default: ONE collective strictly after the backward
            eng.grad_step(x, labels, alpha, head=head)
            dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM, group=self.pg)
            eng.adam_step(lr, grad_scale=1.0 / self.world)
            return
        cut = eng.poff[18]
        total = eng.poff[38]
        if self._comm is None:
            three = os.environ.get("EAE_DP_BUCKETS", "2") == "3"
            self._comm = [eng.dp_stream(0), eng.dp_stream(1) if three else None] if hasattr(eng, "dp_stream") else None
            if self._comm is not None and self._comm[0] is None:
                self._comm = None
        if self._comm is not None:
            # The whole gradient step is enqueued by ONE call (no host gap inside it).  The engine orders its hand-off stream
            # after the completion of gradient tensors 18..37 (decoder side, 3.3 MB): that bucket is exchanged behind that
            # point and overlaps the encoder half of the backward; the encoder-side bucket (2.0 MB) follows after the join.
            # EAE_DP_BUCKETS=3 also hands off tensors 8..17 early, leaving conv1 + conv2 (80 KB) for the end: measured
            # +20 us per extra collective at world size 1, so two buckets are the default.
            c2 = eng.poff[8] if self._comm[1] is not None else 0
            eng.grad_step(x, labels, alpha, head=head)
            with torch.cuda.stream(self._comm[0]):
                h1 = dist.all_reduce(eng.grads[cut:total], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            h2 = None
            if self._comm[1] is not None:
                with torch.cuda.stream(self._comm[1]):
                    h2 = dist.all_reduce(eng.grads[c2:cut], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                dist.all_reduce(eng.grads[0:c2], op=dist.ReduceOp.SUM, group=self.pg)
            else:
                dist.all_reduce(eng.grads[0:cut], op=dist.ReduceOp.SUM, group=self.pg)
            h1.wait()
            if h2 is not None:
                h2.wait()
        else:                                  # split-call form (kept for engines without the hand-off stream)
            eng.grad_step_begin(x, labels, alpha, head=head)
            with torch.cuda.stream(side):
                h1 = dist.all_reduce(eng.grads[cut:total], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            eng.grad_step_end()                # joins every side stream into the current stream
            dist.all_reduce(eng.grads[0:cut], op=dist.ReduceOp.SUM, group=self.pg)
            h1.wait()
        eng.adam_step(lr, grad_scale=1.0 / self.world)


class DPAEStepper:
    """Data-parallel stepper for train.fit_autoencoder (same interface as train.AEStepper): every rank feeds ITS shard of each
    batch; gradients are averaged every step, the epoch scalars (sample-weighted loss sums, counts) are all-reduced ONCE per
    epoch phase, so every rank sees the reference's global epoch means (R.md:656-660, 679-683) and takes the same
    early-stopping decisions."""

    def __init__(self, model, alpha, lr, head=True, max_batch=None, process_group=None, sync_bn=False):
        from .engine import engine_for
        self.model, self.alpha, self.lr, self.head = model, float(alpha), float(lr), head
        self.eng = engine_for(model, max_batch=max_batch)
        self.eng.reset_optimizer()
        self.device = self.eng.device
        self.pg = process_group
        self.trainer = DataParallelTrainer(self.eng, process_group, sync_bn=sync_bn)
        self.trainer.broadcast_parameters()

    def begin(self):
        self.eng.reset_loss()

    def train_step(self, imgs, labels):
        self.trainer.train_step(imgs, labels, self.alpha, self.lr, head=self.head)
        if self.trainer.sync_bn is not None and self.trainer.sync_bn.error is not None:
            raise self.trainer.sync_bn.error

    def eval_step(self, imgs, labels):
        self.eng.forward(imgs, labels=labels, train=False, head=self.head, alpha=self.alpha, want=(), accum=True)

    def end(self):
        acc = self.eng.loss_accum.clone()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.pg)
        a = acc.tolist()
        if hasattr(self.eng, "check_gates"):
            self.eng.check_gates()      # the .tolist() above has synchronised; a timed-out side-stream gate must not go unnoticed
        n = max(a[3], 1.0)
        return a[0] / n, int(a[3])


def fit_autoencoder_dp(train_loader, val_loader, alpha, lr, process_group=None, sync_bn=False, model=None, latent_dim=64,
                       num_classes=10, device="cuda", **kw):
    """train.fit_autoencoder (R.md:619-697) with one process per GPU: `train_loader` / `val_loader` yield THIS rank's shard."""
    from .modules import SupervisedAutoencoder
    from .train import fit_autoencoder, _first_batch_size
    if model is None:
        model = SupervisedAutoencoder(latent_dim=latent_dim, num_classes=num_classes).to(device)
    st = DPAEStepper(model, alpha, lr, head=kw.pop("head", True), process_group=process_group, sync_bn=sync_bn,
                     max_batch=max(_first_batch_size(train_loader), _first_batch_size(val_loader)))
    return fit_autoencoder(train_loader, val_loader, alpha, lr, model=model, stepper=st, **kw)
