"""Host side of the external-MLP engine (R.md:2549-2566): flat arenas + C context, mirroring engine.py."""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib
from ._lib import check
from .engine import _ptr, _stream, _require_gpu

_ENGINES = weakref.WeakKeyDictionary()


class MLPEngine:
    def __init__(self, mlp, max_batch=256):
        self.lib = _lib.load()
        p0 = next(mlp.parameters())
        self.device = p0.device
        _require_gpu(self.device)
        self.input_dim, self.classes, self.max_batch = mlp.input_dim, mlp.num_classes, int(max_batch)
        poff = (C.c_longlong * 11)()
        boff = (C.c_longlong * 5)()
        check(self.lib.eae_mlp_layout(self.input_dim, self.classes, poff, boff))
        self.poff, self.boff = list(poff), list(boff)
        n = self.poff[10]
        dev = self.device
        self.params = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.bn_running = torch.zeros(self.boff[4], dtype=torch.float32, device=dev)
        self.bn_nbt = torch.zeros(2, dtype=torch.int64, device=dev)
        self.stats = torch.zeros(8, dtype=torch.float32, device=dev)
        net = mlp.net
        self._slots = [(net[0].weight, 0), (net[0].bias, 1), (net[1].weight, 2), (net[1].bias, 3), (net[4].weight, 4),
                       (net[4].bias, 5), (net[5].weight, 6), (net[5].bias, 7), (net[7].weight, 8), (net[7].bias, 9)]
        self._bn = [(net[1], 0), (net[5], 1)]
        with torch.no_grad():
            for p, i in self._slots:
                v = self.params[self.poff[i]: self.poff[i] + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = None
            for bn, l in self._bn:
                c = bn.num_features
                rm = self.bn_running[self.boff[2 * l]: self.boff[2 * l] + c]
                rv = self.bn_running[self.boff[2 * l + 1]: self.boff[2 * l + 1] + c]
                rm.copy_(bn.running_mean)
                rv.copy_(bn.running_var)
                self.bn_nbt[l] = bn.num_batches_tracked.to(dev)
                bn._buffers["running_mean"], bn._buffers["running_var"] = rm, rv
                bn._buffers["num_batches_tracked"] = self.bn_nbt[l]
        h = C.c_void_p()
        check(self.lib.eae_mlp_create(self.input_dim, self.classes, self.max_batch, C.byref(h)))
        self.ctx = h
        check(self.lib.eae_mlp_bind(self.ctx, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                    _ptr(self.bn_running), _ptr(self.bn_nbt)))
        self._finalizer = weakref.finalize(self, _destroy, self.lib, self.ctx)
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF

    def attached(self):
        base = self.params.data_ptr()
        return all(p.data_ptr() == base + 4 * self.poff[i] for p, i in self._slots) and \
            self._bn[0][0].running_mean.data_ptr() == self.bn_running.data_ptr()

    def _check(self, x, labels=None):
        if x.device != self.device or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.input_dim:
            raise RuntimeError(f"expected float32 input [B,{self.input_dim}] on {self.device}, got {tuple(x.shape)} {x.dtype} on {x.device}")
        if x.shape[0] > self.max_batch:
            raise RuntimeError(f"batch {x.shape[0]} exceeds the engine's max_batch {self.max_batch}")
        if labels is not None and (labels.dtype != torch.int64 or labels.shape != (x.shape[0],) or labels.device != self.device):
            raise RuntimeError("labels must be int64 [B] on the model's device")
        return x.contiguous(), (None if labels is None else labels.contiguous())

    def forward(self, x, train=False, drop_mask=None):
        x, _ = self._check(x)
        logits = torch.empty((x.shape[0], self.classes), dtype=torch.float32, device=self.device)
        check(self.lib.eae_mlp_forward(self.ctx, _stream(), _ptr(x), x.shape[0], int(train), self.seed, _ptr(drop_mask), _ptr(logits)))
        return logits

    def train_step(self, x, labels, lr, weight_decay=1e-4, drop_mask=None, want_logits=False):
        """One iteration of R.md:2641-2649; loss*B, B and #correct accumulate in self.stats on the device."""
        x, labels = self._check(x, labels)
        logits = torch.empty((x.shape[0], self.classes), dtype=torch.float32, device=self.device) if want_logits else None
        check(self.lib.eae_mlp_train_step(self.ctx, _stream(), _ptr(x), _ptr(labels), x.shape[0], float(lr), float(weight_decay),
                                          self.seed, _ptr(drop_mask), _ptr(logits), _ptr(self.stats)))
        return logits

    def eval_step(self, x, labels, want_logits=False):
        x, labels = self._check(x, labels)
        logits = torch.empty((x.shape[0], self.classes), dtype=torch.float32, device=self.device) if want_logits else None
        check(self.lib.eae_mlp_eval_step(self.ctx, _stream(), _ptr(x), _ptr(labels), x.shape[0], _ptr(logits), _ptr(self.stats)))
        return logits

    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.zero_()
        check(self.lib.eae_mlp_set_adam_step(self.ctx, 0))

    def reset_stats(self):
        self.stats.zero_()

    def read_stats(self):
        """(mean loss, accuracy, n) since reset_stats(); one D2H sync."""
        s = self.stats.tolist()
        n = max(s[1], 1.0)
        return s[0] / n, s[2] / n, int(s[1])


def _destroy(lib, ctx):
    try:
        lib.eae_mlp_destroy(ctx)
    except Exception:
        pass


def mlp_engine_for(mlp, max_batch=None):
    eng = _ENGINES.get(mlp)
    dev = next(mlp.parameters()).device
    _require_gpu(dev)
    want = max_batch or 256
    if eng is not None and (not eng.attached() or eng.device != dev or eng.max_batch < want):
        eng = None
    if eng is None:
        eng = MLPEngine(mlp, max_batch=want)
        _ENGINES[mlp] = eng
    return eng


class _MLPFunction(torch.autograd.Function):
    """Autograd bridge for the reference's own MLP loop (`logits = clf(xb)` ... `loss.backward()`, R.md:2643-2645)."""

    @staticmethod
    def forward(ctx, eng, x, *params):
        x = x.contiguous()
        eng._autograd_calls = getattr(eng, "_autograd_calls", 0) + 1
        seed = (eng.seed + eng._autograd_calls) & 0xFFFFFFFFFFFFFFFF      # fresh dropout mask per forward call
        logits = torch.empty((x.shape[0], eng.classes), dtype=torch.float32, device=eng.device)
        check(eng.lib.eae_mlp_forward(eng.ctx, _stream(), _ptr(x), x.shape[0], 1, seed, None, _ptr(logits)))
        ctx.eng, ctx.seed = eng, seed
        ctx.save_for_backward(x)
        ctx.need = [p.requires_grad for p in params]
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.eng
        (x,) = ctx.saved_tensors
        dlogits = dlogits.to(torch.float32).contiguous()
        check(eng.lib.eae_mlp_backward(eng.ctx, _stream(), _ptr(x), x.shape[0], ctx.seed, None, _ptr(dlogits)))
        grads = [eng.grads[eng.poff[i]: eng.poff[i] + p.numel()].view(p.shape).clone() if need else None
                 for (p, i), need in zip(eng._slots, ctx.need)]
        return (None, None, *grads)


def mlp_forward(module, x):
    eng = mlp_engine_for(module, max_batch=max(256, x.shape[0]))
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        if not module.training:
            raise RuntimeError("differentiating through an eval-mode MLP is not supported by the HIP engine; call clf.train() "
                               "or use torch.no_grad()")
        eng._check(x)
        return _MLPFunction.apply(eng, x, *[p for p, _ in eng._slots])
    return eng.forward(x, train=module.training)
