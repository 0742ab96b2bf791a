"""Which earlier work in the process slows the two-group leg of bench.py's grid measurement?  a = one context from a worker thread, b = four
single-stream contexts from four threads, c = one group of 8 on the default stream, d = two groups of 8 from two threads, e = the same for 10 steps only; e.g.
`python tools/grid_order_probe.py cd`.  Diagnostic tool (GPU box)."""
import os, sys, time, gc, torch
sys.path.insert(0, "/root/repo")
import bench
import eae_amd
from eae_amd.engine import AEEngine, engine_for
from eae_amd import train as T
order = sys.argv[1]
x, y = bench.make_batch(64, torch.device("cuda"), seed=4321)
k = 8
def single(kk, single_stream):
    engs = []
    for i in range(kk):
        torch.manual_seed(100 + i)
        m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
        if single_stream: m._eae_side_streams = -1
        engs.append((m, engine_for(m, max_batch=64)))
    def job_of(e, n):
        def job():
            for _ in range(n): e.train_step(x, y, 35.0, 1e-3)
        return job
    T.run_concurrent([job_of(e, 15) for _, e in engs], kk, static=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); T.run_concurrent([job_of(e, 150) for _, e in engs], kk, static=True); torch.cuda.synchronize()
    print("single", kk, round(kk * 150 * 64 / (time.perf_counter() - t0)), flush=True)
    del engs; gc.collect(); torch.cuda.empty_cache()
def grouped(ng, ss, nsteps=150):
    groups = []
    for g in range(ng):
        es = []
        for i in range(k):
            torch.manual_seed(100 + g * k + i)
            m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
            if ss: m._eae_side_streams = ss
            es.append((m, engine_for(m, max_batch=64)))
        groups.append(es)
    def gjob(es, n):
        engs = [e for _, e in es]; a = ([x] * k, [y] * k, [35.0 + i for i in range(k)], [1e-3] * k)
        def job():
            for _ in range(n): AEEngine.group_train_step(engs, *a)
        return job
    if ng == 1:
        gjob(groups[0], 15)(); torch.cuda.synchronize(); t0 = time.perf_counter(); gjob(groups[0], 150)()
    else:
        T.run_concurrent([gjob(es, 15) for es in groups], ng, static=True); torch.cuda.synchronize()
        t0 = time.perf_counter(); T.run_concurrent([gjob(es, nsteps) for es in groups], ng, static=True)
    torch.cuda.synchronize()
    print("grouped", ng, ss, nsteps, round(ng * k * nsteps * 64 / (time.perf_counter() - t0)), flush=True)
    del groups; gc.collect(); torch.cuda.empty_cache()
for ch in order:
    {"a": lambda: single(1, False), "b": lambda: single(4, True), "c": lambda: grouped(1, 0), "d": lambda: grouped(2, 1), "e": lambda: grouped(2, 1, 10)}[ch]()
