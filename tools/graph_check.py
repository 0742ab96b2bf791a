"""Parameter hash after each of 7 train steps, eager or (EAE_GRAPH=1) graph path, for the product library or a variant (EAE_LIB_PATH).  Diagnostic tool (GPU box)."""
import os, sys, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import golden_util as gu
from helpers import ae_state_np, load_state_np
import ctypes as _C
import eae_amd
from eae_amd import _lib as _L
if os.environ.get("EAE_LIB_PATH"):
    _raw = _C.CDLL(os.environ["EAE_LIB_PATH"])
    for _k in [k for k in _L._PROTOS if not hasattr(_raw, k)]:      # an older variant lacks the newest entry points
        del _L._PROTOS[_k]
from eae_amd.engine import engine_for
x, y = gu.make_images(int(os.environ.get("GC_BATCH", "8")), 100)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
torch.manual_seed(gu.AE_SEED)
m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10)
load_state_np(m, ae_state_np(64)); m = m.cuda()
eng = engine_for(m, max_batch=int(os.environ.get("GC_BATCH", "8")))
hs = []
for s in range(int(os.environ.get('GC_STEPS', '7'))):
    eng.train_step(xd, yd, 35.0, 5e-3)
    torch.cuda.synchronize()
    hs.append("/".join(hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:6] for t in (eng.params, eng.adam_m, eng.adam_v, eng.bn_running, eng.grads, eng.loss_last)))
print(os.environ.get("EAE_GRAPH", "-"), os.environ.get("EAE_LIB_PATH", "new")[-14:], hs, flush=True)
