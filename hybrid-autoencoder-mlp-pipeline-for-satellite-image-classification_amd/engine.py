"""Host side of the HIP engine: flat parameter arenas, the C context, and the step entry points.

The reference keeps every parameter as its own tensor and lets autograd + ``torch.optim.Adam`` walk them
(R.md:624, 646-654).  Here the 38 tensors of ``SupervisedAutoencoder`` live in ONE flat fp32 arena (plus
gradient / Adam-moment arenas of the same layout) so that the fused multi-tensor Adam kernel and the
data-parallel all-reduce see a single buffer; the module's ``nn.Parameter``s are re-pointed to views of the
arena, so ``state_dict()`` / ``load_state_dict()`` / ``.parameters()`` keep working unchanged.

PyTorch is used for device memory, streams and (in ``dp.py``) ``torch.distributed`` only.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib
from ._lib import EaeConfig, EaeStepIO, check

_ENGINES = weakref.WeakKeyDictionary()
_ENGINES_LOCK = __import__("threading").RLock()     # engine_for() may be called from the worker threads of the concurrent grid driver


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(fn):
    """Run an engine method with the engine's device current: the C context allocates its workspace and streams on the current
    device and every launch goes to that device's current stream (a model on cuda:1 with current device 0)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapper


def _require_gpu(device):
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("the autoencoder engine runs on a HIP device only (model.to('cuda')); there is no CPU path")


class AEEngine:
    """Owns the arenas + C context of one SupervisedAutoencoder (or a stand-alone Encoder / Decoder)."""

    def __init__(self, root, max_batch=512, quant=None, side_streams=None):
        from .modules import SupervisedAutoencoder, Encoder, Decoder
        self.quant = {None: 0, "bf16": 0, "fp8": 1, 0: 0, 1: 1}[quant]
        self.lib = _lib.load()
        self.root_ref = weakref.ref(root)
        if isinstance(root, SupervisedAutoencoder):
            enc, dec, cls = root.enc, root.dec, root.classifier
            latent, classes, size = root.latent_dim, root.num_classes, root.enc.image_size
        elif isinstance(root, Encoder):
            enc, dec, cls, latent, classes, size = root, None, None, root.latent_dim, 10, root.image_size
        elif isinstance(root, Decoder):
            enc, dec, cls, latent, classes, size = None, root, None, root.latent_dim, 10, root.image_size
        else:
            raise TypeError(type(root))
        self.latent, self.classes, self.size = latent, classes, size
        p0 = next(root.parameters())
        self.device = p0.device
        _require_gpu(self.device)
        # side_streams: None / 0 = the engine's default (two side streams), 1 / 2 = that many, -1 = none (one stream per context: what
        # several contexts stepped concurrently want, include/eae.h eae_config.side_streams)
        self.side_streams = int(side_streams or 0)
        self.cfg = EaeConfig(latent, classes, size, size, int(max_batch), self.quant, self.side_streams)
        self.max_batch = int(max_batch)
        poff = (C.c_longlong * 39)()
        boff = (C.c_longlong * 15)()
        check(self.lib.eae_ae_layout(C.byref(self.cfg), poff, boff))
        self.poff, self.boff = list(poff), list(boff)
        n = self.poff[38]
        with torch.cuda.device(self.device):
            self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.bn_running = torch.zeros(self.boff[14], dtype=torch.float32, device=self.device)
            self.bn_nbt = torch.zeros(7, dtype=torch.int64, device=self.device)
            self.loss_accum = torch.zeros(8, dtype=torch.float32, device=self.device)
            self.loss_last = torch.zeros(4, dtype=torch.float32, device=self.device)
        # slots: (module attribute path) in named_parameters() order of SupervisedAutoencoder
        self._slots = []      # (param tensor holder, index)
        self._bn_slots = []   # (bn module, l)
        if enc is not None:
            e = enc.encoder
            for i, (ci, bi) in enumerate(((0, 1), (3, 4), (6, 7), (9, 10))):
                self._slots += [(e[ci].weight, 4 * i), (e[ci].bias, 4 * i + 1), (e[bi].weight, 4 * i + 2), (e[bi].bias, 4 * i + 3)]
                self._bn_slots.append((e[bi], i))
            self._slots += [(e[13].weight, 16), (e[13].bias, 17)]
        if dec is not None:
            self._slots += [(dec.decoder_input.weight, 18), (dec.decoder_input.bias, 19)]
            d = dec.decoder
            for i, (di, bi) in enumerate(((1, 2), (4, 5), (7, 8))):
                self._slots += [(d[di].weight, 20 + 4 * i), (d[di].bias, 21 + 4 * i), (d[bi].weight, 22 + 4 * i), (d[bi].bias, 23 + 4 * i)]
                self._bn_slots.append((d[bi], 4 + i))
            self._slots += [(d[10].weight, 32), (d[10].bias, 33)]
        if cls is not None:
            self._slots += [(cls[0].weight, 34), (cls[0].bias, 35), (cls[2].weight, 36), (cls[2].bias, 37)]
        # BatchNorm of absent halves must still be well-defined (gamma=1, running_var=1)
        for l, c in enumerate((32, 64, 128, 256, 128, 64, 32)):
            self.bn_running[self.boff[2 * l + 1]: self.boff[2 * l + 1] + c] = 1.0
            g = (2, 6, 10, 14, 22, 26, 30)[l]
            self.params[self.poff[g]: self.poff[g] + c] = 1.0
        self._adopt()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.eae_create(C.byref(self.cfg), C.byref(h)))
            self.ctx = h
            check(self.lib.eae_bind(self.ctx, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                    _ptr(self.bn_running), _ptr(self.bn_nbt)))
        self._finalizer = weakref.finalize(self, _destroy, self.lib, self.ctx, self.device)
        # one hook per module, reaching whichever engine currently serves it (a rebuilt engine must not keep the old one alive)
        if not getattr(root, "_eae_hooked", False):
            rref = weakref.ref(root)

            def _hook(module, incompatible_keys):
                r = rref()
                eng = _ENGINES.get(r) if r is not None else None
                if eng is not None:
                    eng.params_changed()
            root.register_load_state_dict_post_hook(_hook)
            root._eae_hooked = True

    # ------------------------------------------------------------------ arena management
    def _adopt(self):
        """Copy the module's tensors into the arenas and re-point them to arena views."""
        with torch.no_grad():
            for p, i in self._slots:
                n = p.numel()
                view = self.params[self.poff[i]: self.poff[i] + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = None
            for bn, l in self._bn_slots:
                c = bn.num_features
                rm = self.bn_running[self.boff[2 * l]: self.boff[2 * l] + c]
                rv = self.bn_running[self.boff[2 * l + 1]: self.boff[2 * l + 1] + c]
                rm.copy_(bn.running_mean)
                rv.copy_(bn.running_var)
                self.bn_nbt[l] = bn.num_batches_tracked.to(self.device)
                bn._buffers["running_mean"] = rm
                bn._buffers["running_var"] = rv
                bn._buffers["num_batches_tracked"] = self.bn_nbt[l]

    def attached(self):
        """True while every parameter still aliases the arena (``.to()`` / ``.float()`` may replace them)."""
        base = self.params.data_ptr()
        for p, i in self._slots:
            if p.data_ptr() != base + 4 * self.poff[i]:
                return False
        for bn, l in self._bn_slots:
            if bn.running_mean.data_ptr() != self.bn_running.data_ptr() + 4 * self.boff[2 * l]:
                return False
        return True

    def params_changed(self):
        check(self.lib.eae_params_changed(self.ctx))

    def set_graph(self, on=True):
        """Replay train_step from a captured hipGraph (one host call per step) instead of ~70 eager launches: for host-bound use,
        i.e. several small configurations stepped concurrently (train.run_concurrent)."""
        check(self.lib.eae_set_graph(self.ctx, int(bool(on))))

    def gate_timeouts(self):
        """0, or the progress value a side-stream gate gave up waiting for (diagnostic; synchronises the device)."""
        with torch.cuda.device(self.device):
            return int(self.lib.eae_gate_timeouts(self.ctx))

    def check_gates(self, sync=True):
        """Raise if a side-stream gate of this engine has ever timed out (sticky; the optimizer kernels have refused to update since,
        and the loss scalars of those steps are NaN).  sync=True synchronises the device; sync=False only reads the word (for callers
        that have just synchronised the stream they step on, e.g. after read_loss(): the concurrent grid driver)."""
        if sync:
            v = self.gate_timeouts()
        else:
            with torch.cuda.device(self.device):
                v = int(self.lib.eae_gate_timeouts_nosync(self.ctx))
        if v:
            raise _lib.EaeError(f"a side-stream gate timed out waiting for progress value {v}: weight gradients of that step may come "
                                "from stale activations; no parameter update has been applied since (raise EAE_GATE_TIMEOUT_MS, or "
                                "EAE_FORK_EVENTS=1 for event hand-overs; clear with engine.clear_gate_timeouts())")

    def clear_gate_timeouts(self):
        with torch.cuda.device(self.device):
            check(self.lib.eae_gate_timeouts_clear(self.ctx))

    def generation(self):
        """Id of the most recent forward (eae_ae_backward only differentiates the resident one)."""
        return int(self.lib.eae_forward_generation(self.ctx))

    def expose_grads(self):
        """Make ``p.grad`` of every parameter a view of the gradient arena (after a fused grad step)."""
        for p, i in self._slots:
            p.grad = self.grads[self.poff[i]: self.poff[i] + p.numel()].view(p.shape)

    def grad_view(self, i, shape):
        n = 1
        for s in shape:
            n *= s
        return self.grads[self.poff[i]: self.poff[i] + n].view(shape)

    # ------------------------------------------------------------------ steps
    def _io(self, x, labels, train, head, alpha, x_hat=None, logits=None, z=None, accum=True):
        if x.device != self.device or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3 or \
                x.shape[2] != self.size or x.shape[3] != self.size:
            raise RuntimeError(f"expected float32 input [B,3,{self.size},{self.size}] on {self.device}, got "
                               f"{tuple(x.shape)} {x.dtype} on {x.device}")
        if not x.is_contiguous():
            x = x.contiguous()
        b = x.shape[0]
        if b > self.max_batch:
            raise RuntimeError(f"batch {b} exceeds the engine's max_batch {self.max_batch}")
        if labels is not None:
            if labels.dtype != torch.int64 or labels.shape != (b,) or labels.device != self.device:
                raise RuntimeError("labels must be int64 [B] on the model's device")
            labels = labels.contiguous()
        io = EaeStepIO(_ptr(x), _ptr(labels), b, int(train), int(head), float(alpha), _ptr(x_hat), _ptr(logits), _ptr(z),
                       _ptr(self.loss_accum) if accum else None, _ptr(self.loss_last))
        return io, (x, labels)

    @_on_device
    def forward(self, x, labels=None, train=False, head=True, alpha=1.0, want=("x_hat", "logits", "z"), accum=False):
        b = x.shape[0]
        x_hat = torch.empty((b, 3, self.size, self.size), dtype=torch.float32, device=self.device) if "x_hat" in want else None
        logits = torch.empty((b, self.classes), dtype=torch.float32, device=self.device) if ("logits" in want and head) else None
        z = torch.empty((b, self.latent), dtype=torch.float32, device=self.device) if "z" in want else None
        io, keep = self._io(x, labels, train, head, alpha, x_hat, logits, z, accum)
        check(self.lib.eae_ae_forward(self.ctx, _stream(), C.byref(io)))
        return x_hat, logits, z

    @_on_device
    def grad_step(self, x, labels, alpha, head=True, x_hat=None):
        io, keep = self._io(x, labels, True, head, alpha, x_hat)
        check(self.lib.eae_ae_grad_step(self.ctx, _stream(), C.byref(io)))

    @_on_device
    def adam_step(self, lr, weight_decay=0.0, grad_scale=1.0):
        check(self.lib.eae_adam_step_scaled(self.ctx, _stream(), float(lr), float(weight_decay), float(grad_scale)))

    @_on_device
    def grad_step_begin(self, x, labels, alpha, head=True):
        """forward + loss + backward of classifier / decoder / dec.fc: gradient tensors 18..37 complete behind the side stream"""
        io, keep = self._io(x, labels, True, head, alpha)
        self._keep = keep
        check(self.lib.eae_ae_grad_step_begin(self.ctx, _stream(), C.byref(io)))

    @_on_device
    def grad_step_end(self):
        check(self.lib.eae_ae_grad_step_end(self.ctx, _stream()))
        self._keep = None

    def side_stream(self):
        """The engine's side stream as a torch ExternalStream (None when side-stream concurrency is disabled)."""
        h = self.lib.eae_side_stream(self.ctx)
        return torch.cuda.ExternalStream(h, device=self.device) if h else None

    def dp_stream(self, which=0):
        """Engine stream that after every grad_step() is ordered after gradient tensors 18..37 (which=0) or 8..17 (which=1),
        as a torch ExternalStream."""
        h = self.lib.eae_dp_stream(self.ctx, int(which))
        return torch.cuda.ExternalStream(h, device=self.device) if h else None

    @_on_device
    def train_step(self, x, labels, alpha, lr, head=True, x_hat=None):
        """One iteration of the reference's batch loop (R.md:646-657); the loss is accumulated on the device."""
        io, keep = self._io(x, labels, True, head, alpha, x_hat)
        check(self.lib.eae_ae_train_step(self.ctx, _stream(), C.byref(io), float(lr)))

    @staticmethod
    def _group_ios(engines, xs, labels, alphas, train, head, accum=True):
        n = len(engines)
        e0 = engines[0]
        if any(e.device != e0.device for e in engines):
            raise RuntimeError("group call: the engines must live on one device")
        ios = (EaeStepIO * n)()
        keep = []
        for k, e in enumerate(engines):
            io, kp = e._io(xs[k], labels[k], train, head, alphas[k], accum=accum)
            ios[k] = io
            keep.append(kp)
        return ios, (C.c_void_p * n)(*[e.ctx for e in engines]), keep

    @staticmethod
    def group_train_step(engines, xs, labels, alphas, lrs, head=True, geometry_mult=0):
        """One iteration of the batch loop for SEVERAL configurations of the grid at once (R.md:599-711: same architecture, own
        alpha / lr / parameters / batches): include/eae.h eae_group_train_step -- one sequence of grouped launches instead of one per
        engine.  engines: same shape, same device; xs / labels: one batch each (same batch size)."""
        n = len(engines)
        ios, ctxs, keep = AEEngine._group_ios(engines, xs, labels, alphas, True, head)
        lr_arr = (C.c_float * n)(*[float(v) for v in lrs])
        e0 = engines[0]
        with torch.cuda.device(e0.device):
            check(e0.lib.eae_group_train_step(ctxs, n, int(geometry_mult), _stream(), ios, lr_arr))

    @staticmethod
    def group_eval_step(engines, xs, labels, alphas, head=True, geometry_mult=0):
        """The validation pass of the same configurations (R.md:670-682): eval-mode forward + loss, accumulated on the device."""
        n = len(engines)
        ios, ctxs, keep = AEEngine._group_ios(engines, xs, labels, alphas, False, head)
        e0 = engines[0]
        with torch.cuda.device(e0.device):
            check(e0.lib.eae_group_forward(ctxs, n, int(geometry_mult), _stream(), ios))

    @_on_device
    def encoder(self, x, train=False):
        b = x.shape[0]
        z = torch.empty((b, self.latent), dtype=torch.float32, device=self.device)
        io, keep = self._io(x, None, train, False, 1.0)
        check(self.lib.eae_encoder_forward(self.ctx, _stream(), _ptr(keep[0]), b, int(train), _ptr(z)))
        return z

    @_on_device
    def decoder(self, z, train=False):
        if z.device != self.device or z.dtype != torch.float32 or z.dim() != 2 or z.shape[1] != self.latent:
            raise RuntimeError(f"expected float32 latent [B,{self.latent}] on {self.device}")
        z = z.contiguous()
        b = z.shape[0]
        if b > self.max_batch:
            raise RuntimeError(f"batch {b} exceeds the engine's max_batch {self.max_batch}")
        x_hat = torch.empty((b, 3, self.size, self.size), dtype=torch.float32, device=self.device)
        check(self.lib.eae_decoder_forward(self.ctx, _stream(), _ptr(z), b, int(train), _ptr(x_hat)))
        return x_hat

    @_on_device
    def fp8_calibrate(self, x, labels, alpha, head=True, iters=7):
        """fp8 variant: settle the delayed scales on one batch before the first real step (include/eae.h, eae_fp8_calibrate)."""
        io, keep = self._io(x, labels, True, head, alpha, None)
        check(self.lib.eae_fp8_calibrate(self.ctx, _stream(), C.byref(io), int(iters)))

    def fp8_scales(self):
        """dict of the current per-tensor scales of the fp8 variant: 'act', 'grad', 'w' -> 6 floats (conv2..conv4, deconv1..deconv3)."""
        out = (C.c_float * 18)()
        check(self.lib.eae_fp8_scales(self.ctx, out))
        v = list(out)
        return {"act": v[0:6], "grad": v[6:12], "w": v[12:18]}

    @_on_device
    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.zero_()
        check(self.lib.eae_set_adam_step(self.ctx, 0))

    def reset_loss(self):
        self.loss_accum.zero_()

    def read_loss(self):
        """(mean loss, mean mse, mean ce, n samples, n correct) accumulated since reset_loss(); one D2H sync."""
        a = self.loss_accum.tolist()
        n = max(a[3], 1.0)
        return a[0] / n, a[1] / n, a[2] / n, int(a[3]), int(a[4])


def _destroy(lib, ctx, device=None):
    try:
        if device is not None:
            with torch.cuda.device(device):
                lib.eae_destroy(ctx)
        else:
            lib.eae_destroy(ctx)
    except Exception:
        pass


_PARENTS = weakref.WeakKeyDictionary()     # Encoder / Decoder instance -> weakref(SupervisedAutoencoder that owns it)


def register_children(parent, children):
    for ch in children:
        _PARENTS[ch] = weakref.ref(parent)


def _root_of(module):
    parent = _PARENTS.get(module)
    if parent is not None:
        p = parent()
        if p is not None:
            return p
    return module


def engine_for(module, max_batch=None, quant=None):
    """Engine of the SupervisedAutoencoder that owns `module` (or of a stand-alone Encoder / Decoder).  quant="fp8": BASELINE config
    5's variant (fp8 operands for the GEMMs of the six 3x3 layers, include/eae.h); it is a property of the engine, chosen when the
    engine of a module is first built (`module._eae_quant = "fp8"` before the first use does the same)."""
    with _ENGINES_LOCK:
        return _engine_for_locked(module, max_batch, quant)


def _engine_for_locked(module, max_batch, quant):
    root = _root_of(module)
    eng = _ENGINES.get(root)
    quant = quant if quant is not None else getattr(root, "_eae_quant", None)
    if eng is not None and quant is not None and eng.quant != {"bf16": 0, "fp8": 1, 0: 0, 1: 1}[quant]:
        raise RuntimeError("this module already has an engine with a different `quant`; the operand format is fixed when the engine is "
                           "first built (set module._eae_quant before the first forward, or pass quant= to the first engine_for call)")
    dev = next(root.parameters()).device
    _require_gpu(dev)
    want_mb = max_batch or getattr(root, "_eae_max_batch", 512)
    old = None
    if eng is not None and (not eng.attached() or eng.device != dev):
        eng = None     # parameters were moved / re-created: rebuild the arenas from the module's current tensors
    elif eng is not None and eng.max_batch < want_mb:
        old, eng = eng, None     # same parameters, bigger workspace: the optimizer state moves to the new engine
    if eng is None:
        eng = AEEngine(root, max_batch=want_mb, quant=quant if old is None else old.quant,
                       side_streams=getattr(root, "_eae_side_streams", None) if old is None else old.side_streams)
        if old is not None:
            with torch.no_grad():
                eng.adam_m.copy_(old.adam_m)
                eng.adam_v.copy_(old.adam_v)
                eng.loss_accum.copy_(old.loss_accum)
            check(eng.lib.eae_set_adam_step(eng.ctx, old.lib.eae_get_adam_step(old.ctx)))
        _ENGINES[root] = eng
    return eng


class _AEFunction(torch.autograd.Function):
    """Autograd bridge for hand-written loops (the reference's own: `x_hat, logits, _ = model(imgs)` ... `loss.backward()`,
    R.md:647-653): forward = eae_ae_forward, backward = eae_ae_backward with the gradients of (x_hat, logits, z).  The
    parameter gradients are returned to autograd (which accumulates them into `.grad` like for any torch module)."""

    @staticmethod
    def forward(ctx, eng, train, x, *params):
        x = x.contiguous()          # the backward reads the batch again (conv1 weight gradient): keep OUR copy alive, not the caller's
        x_hat, logits, z = eng.forward(x, train=train, head=True)
        ctx.eng = eng
        ctx.gen = eng.generation()
        ctx.save_for_backward(x, x_hat)
        ctx.nparams = len(params)
        ctx.need = [p.requires_grad for p in params]
        return x_hat, logits, z

    @staticmethod
    def backward(ctx, dx_hat, dlogits, dz):
        eng = ctx.eng
        x, x_hat = ctx.saved_tensors

        def prep(t):
            return None if t is None else t.to(dtype=torch.float32).contiguous()

        dx_hat = prep(dx_hat)
        if dx_hat is None:
            dx_hat = torch.zeros_like(x_hat)
        dlogits, dz = prep(dlogits), prep(dz)
        with torch.cuda.device(eng.device):
            check(eng.lib.eae_ae_backward(eng.ctx, _stream(), ctx.gen, _ptr(x), _ptr(x_hat), _ptr(dx_hat), _ptr(dlogits), _ptr(dz)))
        grads = []
        for (p, i), need in zip(eng._slots, ctx.need):
            grads.append(eng.grads[eng.poff[i]: eng.poff[i] + p.numel()].view(p.shape).clone() if need else None)
        return (None, None, None, *grads)


class _HalfFunction(torch.autograd.Function):
    """Autograd through a stand-alone Encoder (half = "enc": z = enc(x)) or Decoder (half = "dec": x_hat = dec(z)), e.g. a plain
    autoencoder composed by hand, `x_hat = dec(enc(x))`.  forward = eae_encoder_forward / eae_decoder_forward, backward =
    eae_encoder_backward / eae_decoder_backward; only the parameters of that half receive gradients."""

    @staticmethod
    def forward(ctx, eng, half, train, inp, *params):
        inp = inp.contiguous()
        out = eng.encoder(inp, train=train) if half == "enc" else eng.decoder(inp, train=train)
        ctx.eng, ctx.half, ctx.gen = eng, half, eng.generation()
        if half == "enc":
            eng._pending_enc_gen = ctx.gen       # an encoder forward that still awaits its backward is resident in this engine
        ctx.save_for_backward(inp, out)
        ctx.slots = [(p, i) for p, i in eng._slots if (i < 18 if half == "enc" else 18 <= i < 34)]
        ctx.need = [p.requires_grad for p in params]
        ctx.need_input = inp.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        inp, out = ctx.saved_tensors
        dout = dout.to(dtype=torch.float32).contiguous()
        din = None
        with torch.cuda.device(eng.device):
            if ctx.half == "enc":
                check(eng.lib.eae_encoder_backward(eng.ctx, _stream(), ctx.gen, _ptr(inp), _ptr(dout)))     # (no dL/dx: images are leaves)
                eng._pending_enc_gen = None
            else:
                din = torch.empty_like(inp) if ctx.need_input else None
                check(eng.lib.eae_decoder_backward(eng.ctx, _stream(), ctx.gen, _ptr(out), _ptr(dout), _ptr(din)))
        grads = [eng.grads[eng.poff[i]: eng.poff[i] + p.numel()].view(p.shape).clone() if need else None
                 for (p, i), need in zip(ctx.slots, ctx.need)]
        return (None, None, None, din, *grads)


def _half_forward(module, inp, half):
    eng = engine_for(module)
    slots = [p for p, i in eng._slots if (i < 18 if half == "enc" else 18 <= i < 34)]
    grad_mode = torch.is_grad_enabled()
    if grad_mode and half == "enc" and inp.requires_grad:
        # the engine computes no gradient w.r.t. the images (conv1 has no backward-data: the reference's inputs are leaves, R.md:643-647);
        # silently returning None would starve a differentiable stage in front of the encoder
        raise RuntimeError("Encoder.forward: the input requires grad, but the HIP engine does not compute dL/dx (images are leaves of the "
                           "reference's graph); detach the input, or keep learnable pre-processing outside the differentiated path")
    if grad_mode and half == "dec" and getattr(eng, "_pending_enc_gen", None) == eng.generation():
        # model.dec(model.enc(x)) on the two halves of ONE SupervisedAutoencoder: both halves share one engine workspace and the
        # decoder forward would replace the resident encoder forward that still awaits its backward
        raise RuntimeError("Decoder.forward on the decoder of the SupervisedAutoencoder whose encoder forward is awaiting its backward: the "
                           "two halves share one engine; call model(x) (x_hat, logits, z = model(x)), or use stand-alone Encoder / Decoder "
                           "modules for a hand-composed dec(enc(x))")
    eng.params_changed()           # (after the guards: it drops whatever forward is resident in the engine)
    if grad_mode and (inp.requires_grad or any(p.requires_grad for p in slots)):
        return _HalfFunction.apply(eng, half, bool(module.training), inp, *slots)
    return eng.encoder(inp, train=module.training) if half == "enc" else eng.decoder(inp, train=module.training)


def autoencoder_forward(module, x):
    eng = engine_for(module)
    eng.params_changed()
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        # (eval mode: BatchNorm uses the running statistics in the forward and is differentiated as the per-channel affine map it then is)
        return _AEFunction.apply(eng, bool(module.training), x, *[p for p, _ in eng._slots])
    x_hat, logits, z = eng.forward(x, train=module.training, head=True)
    return x_hat, logits, z


def encoder_forward(module, x):
    return _half_forward(module, x, "enc")


def decoder_forward(module, z):
    return _half_forward(module, z, "dec")
