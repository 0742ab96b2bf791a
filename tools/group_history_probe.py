import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import eae_amd
from eae_amd.engine import AEEngine, engine_for
from eae_amd import train as T
mode = sys.argv[1]
x = torch.rand((64, 3, 64, 64), device="cuda"); y = torch.randint(0, 10, (64,), device="cuda")
def build(n, single):
    out = []
    for i in range(n):
        torch.manual_seed(100 + i)
        m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
        if single: m._eae_side_streams = -1
        out.append((m, engine_for(m, max_batch=64)))
    return out
def grouped(tag):
    engs = build(8, False); es = [e for _, e in engs]
    a = ([x] * 8, [y] * 8, [35.0] * 8, [1e-3] * 8)
    for _ in range(15): AEEngine.group_train_step(es, *a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(150): AEEngine.group_train_step(es, *a)
    torch.cuda.synchronize(); print(tag, "grouped ms", round((time.perf_counter() - t0) / 150 * 1e3, 4), flush=True)
def conc(kk, single):
    engs = build(kk, single)
    def job_of(e, n):
        def job():
            for _ in range(n): e.train_step(x, y, 35.0, 1e-3)
        return job
    T.run_concurrent([job_of(e, 165) for _, e in engs], kk); torch.cuda.synchronize()
if mode == "a": grouped("fresh")
if mode == "b": conc(1, False); grouped("after k1 threads")
if mode == "c": conc(4, True); grouped("after 4 single-stream ctx")
if mode == "d":
    e = build(1, False)
    for _ in range(165): e[0][1].train_step(x, y, 35.0, 1e-3)
    torch.cuda.synchronize(); del e; grouped("after k1 main thread")
if mode == "e":
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): grouped("on a torch stream")
if mode == "f": conc(1, False); conc(4, True); grouped("after k1 + 4 ctx")
if mode == "g":
    import bench
    r = bench.grid_b64_leg()
    print("bench leg", r["grouped"], flush=True)
if mode == "h":
    import gc
    conc(1, False); conc(4, True); gc.collect(); torch.cuda.empty_cache(); grouped("after k1 + 4 ctx + gc")
