"""Data-parallel training: one process per GPU, gradients averaged with RCCL over xGMI (torch.distributed 'nccl').

The reference has no parallelism of any kind; this is new work required by BASELINE.json's north_star.
Every rank holds a full replica (1.3 M parameters) and a shard of the global minibatch; per step the flat fp32 gradient
arena (5.26 MB) is all-reduced in a few large buckets and divided by the world size, then the fused Adam runs on every
rank (identical updates keep the replicas in sync).  BatchNorm uses per-rank batch statistics (DDP semantics).

Two exchanges: the ENGINE's own RCCL communicator (``eae_dp_init`` / ``eae_ae_dp_train_step``: forward, backward, all-reduce and
Adam enqueued by one C call; the default on the nccl backend, ``EAE_DP_NATIVE=0`` switches it off), and ``torch.distributed`` (gloo
in the CPU tests, engines without a C context, the fall-back when the engine's communicator cannot be brought up on EVERY rank).

Default on both: ONE all-reduce of the whole arena strictly after the backward.  ``EAE_DP_OVERLAP=1`` hands the decoder-side bucket
to RCCL while the encoder half of the backward is still running; it stays opt-in until it has run on more than
one rank (one communicator driven from two streams, the hand-off covering tensors written on the side streams), and it is switched
off with ``sync_bn=True`` (SyncBN's collectives of the encoder backward would run beside the engine's bucket on another communicator).

The decision "no update this step" (a timed-out side-stream gate, a non-finite BatchNorm statistic) is taken for all ranks together:
the flag word is max-reduced with the gradients, and ``DPAEStepper.end`` all-reduces the gate word with the epoch scalars so that every
rank raises in the same place.

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

    def _agree(self, ok):
        """Every rank learns whether EVERY rank is ok: ONE all-reduce(MIN) of a flag, issued by every rank whatever failed locally."""
        f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.eng.device)
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=self.pg)
        return int(f.item()) == 1

    def _init_native_checked(self):
        """Bring up the engine's communicator and prove it on a known vector before it is trusted with real gradients.  The sequence
        of torch collectives is IDENTICAL on every rank whatever fails locally (a rank that raised in front of a broadcast
        while its peers sat in it, or good ranks spinning in the self-check all-reduce a failed rank never entered, hung the group):
          0. agree: does every rank already own a communicator of this size?  (all or nothing)
          1. every rank checks locally that it can reach RCCL (eae_dp_unique_id: dlopen + symbols; cannot block); rank 0's id is
             broadcast -- zeroed if rank 0 failed -- and the ranks agree on "everybody can";
          2. only then every rank calls eae_dp_init (a rendezvous inside RCCL, time-bounded by EAE_DP_INIT_TIMEOUT_S); agree;
          3. only if every rank joined: all-reduce of rank + 1 over 64 gradient words; agree on the result.
        Any "no" sends ALL ranks to the torch.distributed exchange together."""
        import ctypes as C
        from ._lib import check
        from .engine import _stream
        eng = self.eng
        why = []

        def attempt(fn, what):
            try:
                fn()
                return True
            except Exception as e:      # a missing librccl symbol, a refused communicator, ...
                why.append(f"{what}: {type(e).__name__}: {e}")
                return False

        def on_dev(fn):
            if eng.device.type == "cuda":
                with torch.cuda.device(eng.device):
                    return fn()
            return fn()

        have = [False]
        attempt(lambda: have.__setitem__(0, int(eng.lib.eae_dp_world(eng.ctx)) == self.world), "eae_dp_world")
        joined = self._agree(have[0])
        if not joined:
            idbuf = (C.c_ubyte * 128)()

            def preflight():
                if int(eng.lib.eae_dp_world(eng.ctx)) != 0:          # a communicator of another size (or a half-agreed one): drop it
                    on_dev(lambda: eng.lib.eae_dp_destroy(eng.ctx))
                check(eng.lib.eae_dp_unique_id(idbuf))
            can = attempt(preflight, "eae_dp_unique_id")
            t = torch.zeros(128, dtype=torch.uint8, device=eng.device)
            if self.rank == 0 and can:
                t.copy_(torch.tensor(list(idbuf), dtype=torch.uint8))
            src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
            dist.broadcast(t, src=src, group=self.pg)               # ALWAYS (zeros when rank 0 could not draw an id)
            if self._agree(can):
                raw = bytes(t.cpu().tolist())
                ok = attempt(lambda: on_dev(lambda: check(eng.lib.eae_dp_init(eng.ctx, self.rank, self.world, raw))), "eae_dp_init")
                joined = self._agree(ok)
        proven = False
        if joined:
            def selfcheck():
                keep = eng.grads[:64].clone()
                eng.grads[:64] = float(self.rank + 1)
                rc = on_dev(lambda: int(eng.lib.eae_dp_allreduce_bucket(eng.ctx, _stream() if eng.device.type == "cuda" else None,
                                                                        C.c_longlong(0), C.c_longlong(64))))
                if eng.device.type == "cuda":
                    torch.cuda.synchronize(eng.device)
                want = self.world * (self.world + 1) / 2.0
                got = float(eng.grads[0])
                good = rc == 0 and bool((eng.grads[:64] == want).all())
                eng.grads[:64] = keep
                if not good:
                    raise RuntimeError(f"self-check of the engine's all-reduce failed (rc {rc}, got {got}, want {want})")
            proven = self._agree(attempt(selfcheck, "self-check"))
        self.native = bool(joined and proven)
        if not self.native:
            if why:
                print(f"[eae dp] rank {self.rank}: engine-owned RCCL exchange unavailable ({'; '.join(why)}); using torch.distributed",
                      file=sys.stderr, flush=True)
            try:
                on_dev(lambda: eng.lib.eae_dp_destroy(eng.ctx))
            except Exception:
                pass

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
            # opt-in, and never beside SyncBN's collectives (module docstring)
            overlap = 1 if (os.environ.get("EAE_DP_OVERLAP", "0") == "1" and self.sync_bn is None) else 0
            with torch.cuda.device(eng.device):
                check(eng.lib.eae_ae_dp_train_step(eng.ctx, _stream(), C.byref(io), float(lr), overlap))
            return
        side = eng.side_stream() if hasattr(eng, "grad_step_begin") and hasattr(eng, "side_stream") else None
        if side is None:                       # engines without the split API (CPU stand-in of the gloo test)
            eng.grad_step(x, labels, alpha, head=head)
            self.allreduce_gradients()
            eng.adam_step(lr)
            return
        if os.environ.get("EAE_DP_OVERLAP", "0") != "1" or self.sync_bn is not None:     # default: ONE collective strictly after the backward
            eng.grad_step(x, labels, alpha, head=head)
            dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM, group=self.pg)
            self._adam_together(lr)
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
        self._adam_together(lr)

    def _adam_together(self, lr):
        """Adam with the 1/world scale folded in; the refusal switches (gate time-out, non-finite statistics) max-reduced over the ranks
        first, so that either every replica updates or none does (engines with the C entry points; others: plain adam_step)."""
        eng = self.eng
        if not (hasattr(eng, "ctx") and hasattr(eng.lib, "eae_adam_step_dp")):
            eng.adam_step(lr, grad_scale=1.0 / self.world)
            return
        import ctypes as C
        from ._lib import check
        from .engine import _stream
        if not hasattr(self, "_bad"):
            self._bad = torch.zeros(2, dtype=torch.int32, device=eng.device)      # (stale, diverged)
        with torch.cuda.device(eng.device):
            check(eng.lib.eae_dp_local_bad(eng.ctx, _stream(), C.c_void_p(self._bad.data_ptr())))
            dist.all_reduce(self._bad, op=dist.ReduceOp.MAX, group=self.pg)
            check(eng.lib.eae_adam_step_dp(eng.ctx, _stream(), float(lr), 0.0, 1.0 / self.world, C.c_void_p(self._bad.data_ptr())))


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
        # the gate word travels WITH the epoch scalars: a rank whose side-stream gate timed out must not raise alone while its peers
        # walk into their next collective -- every rank sees the sum and raises in the same place
        gates = float(bool(self.eng.gate_timeouts())) if hasattr(self.eng, "gate_timeouts") else 0.0      # (synchronises; once per epoch phase)
        acc = torch.cat([self.eng.loss_accum.detach().to(torch.float32).flatten(),
                         torch.tensor([1.0 if gates > 0 else 0.0], dtype=torch.float32, device=self.eng.loss_accum.device)])
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.pg)
        a = acc.tolist()
        if hasattr(self.eng, "check_gates"):
            self.eng.check_gates()      # the .tolist() above has synchronised: this rank's own time-out, with its details
        if a[-1] > 0:
            raise RuntimeError(f"{int(a[-1])} data-parallel rank(s) reported a timed-out side-stream gate: the step's gradients may come from "
                               "stale activations; no replica was updated (see eae_gate_timeouts, EAE_GATE_TIMEOUT_MS)")
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
