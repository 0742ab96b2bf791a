"""Host-side cost of enqueueing one train step vs its GPU duration (is the step launch-bound?).  usage: python tools/hosttime.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eae_amd  # noqa: E402
from eae_amd.engine import engine_for  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
m = eae_amd.SupervisedAutoencoder(64, 10).cuda().train()
eng = engine_for(m, max_batch=B)
x = torch.rand(B, 3, 64, 64, device="cuda")
y = torch.randint(0, 10, (B,), device="cuda")
for _ in range(30):
    eng.train_step(x, y, 35.0, 1e-3)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    eng.train_step(x, y, 35.0, 1e-3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e6 * (t1 - t0) / N:.1f} us/step, wall {1e6 * (t2 - t0) / N:.1f} us/step "
      f"(queue drained {1e3 * (t2 - t1):.2f} ms after the last enqueue)")

# the same step through the C entry point alone (argument struct built once): what the Python wrapper costs on top
import ctypes as C  # noqa: E402
io, keep = eng._io(x, y, True, True, 35.0, None)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lr = C.c_float(1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    eng.lib.eae_ae_train_step(eng.ctx, st, C.byref(io), lr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: C call alone: host enqueue {1e6 * (t1 - t0) / N:.1f} us/step, wall {1e6 * (t2 - t0) / N:.1f} us/step")
