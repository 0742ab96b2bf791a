"""Grid-level execution (SURVEY.md 8f N2, R.md:599-711): K configurations of the reference's alpha x lr grid trained CONCURRENTLY on
one GPU -- one engine context per configuration, each on its own stream, stepped from its own host thread -- give, for every
configuration, bitwise the curves and final weights of the same configuration trained alone."""
import threading

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _loaders():
    tr = [(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in (gu.make_images(b, 1300 + i) for i, b in enumerate((64, 64, 48)))]
    va = [(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in (gu.make_images(b, 1400 + i) for i, b in enumerate((64, 56)))]
    return tr, va


def test_concurrent_grid_configs_are_bitwise_the_sequential_ones(tmp_path):
    import eae_amd
    from eae_amd import train as T
    tr, va = _loaders()
    lock = threading.Lock()
    store = {}

    def fit(tag):
        def fit_fn(train_loader, val_loader, alpha, lr, **kw):
            with lock:                                   # the module initialisers draw from torch's GLOBAL generator
                torch.manual_seed(1000 + int(alpha) * 7 + int(lr * 1e4))
                model = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda()
            kw.pop("model", None)
            r = T.fit_autoencoder(train_loader, val_loader, alpha, lr, model=model, **kw)
            store[(tag, alpha, lr)] = (list(r["train_curve"]), list(r["val_curve"]),
                                       {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()})
            return r
        return fit_fn

    kw = dict(alpha_values=(20, 35), lr_values=(1e-3, 5e-3), num_epochs=2, patience=15, verbose=False)
    seq = T.grid_search_autoencoder(tr, va, out_dir=str(tmp_path / "seq"), fit_fn=fit("seq"), concurrent=1, **kw)
    con = T.grid_search_autoencoder(tr, va, out_dir=str(tmp_path / "con"), fit_fn=fit("con"), concurrent=4, **kw)
    assert seq["results"] == con["results"] and (seq["best_alpha"], seq["best_lr"]) == (con["best_alpha"], con["best_lr"])
    for alpha in (20, 35):
        for lr in (1e-3, 5e-3):
            a, b = store[("seq", alpha, lr)], store[("con", alpha, lr)]
            assert a[0] == b[0] and a[1] == b[1], (alpha, lr, a[:2], b[:2])
            assert np.isfinite(a[0]).all() and a[0][-1] < a[0][0]
            for k in a[2]:
                assert np.array_equal(a[2][k], b[2][k]), (alpha, lr, k)
    sa, sb = torch.load(seq["best_path"]), torch.load(con["best_path"])
    assert all(torch.equal(sa[k], sb[k]) for k in sa)


def test_concurrent_grid_with_the_default_fit_fn_is_seeded_like_the_sequential_one(tmp_path):
    """The DEFAULT fit_fn under torch.manual_seed: the concurrent grid builds its models in grid order on the calling
    thread, so every configuration starts from the parameters the sequential grid draws for it -- same validation losses bit for
    bit, same winner, same saved weights (list loaders: no shuffling that would draw from the global generator)."""
    from eae_amd import train as T
    tr, va = _loaders()
    kw = dict(alpha_values=(20, 35), lr_values=(1e-3, 5e-3), num_epochs=2, patience=15, verbose=False)
    torch.manual_seed(4321)
    seq = T.grid_search_autoencoder(tr, va, out_dir=str(tmp_path / "seq"), concurrent=1, **kw)
    for k in (2, 4):
        torch.manual_seed(4321)
        con = T.grid_search_autoencoder(tr, va, out_dir=str(tmp_path / f"con{k}"), concurrent=k, **kw)
        assert seq["results"] == con["results"], (k, seq["results"], con["results"])
        assert (seq["best_alpha"], seq["best_lr"]) == (con["best_alpha"], con["best_lr"])
        sa, sb = torch.load(seq["best_path"]), torch.load(con["best_path"])
        assert all(torch.equal(sa[key], sb[key]) for key in sa)


def test_concurrent_steps_from_threads_scale_and_stay_deterministic():
    """K engines stepped from K threads (train.run_concurrent): same parameters as the same steps issued from one thread."""
    import eae_amd
    from eae_amd import train as T
    from eae_amd.engine import engine_for
    x, y = gu.make_images(64, 77)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()

    def make(i, single_stream=False):
        torch.manual_seed(50 + i)
        m = eae_amd.SupervisedAutoencoder(64).cuda().train()
        if single_stream:
            m._eae_side_streams = -1        # eae_config.side_streams = -1: what grid_search_autoencoder(concurrent >= 3) builds
        return m, engine_for(m, max_batch=64)

    ref = []
    for i in range(4):
        m, e = make(i)
        for _ in range(5):
            e.train_step(xd, yd, 35.0, 1e-3)
        torch.cuda.synchronize()
        ref.append(e.params.clone())
    # the concurrent contexts keep every kernel on their ONE stream (four hardware queues: four contexts side by side); the reference
    # runs above used the default three streams per context -- same bits either way
    pairs = [make(i, single_stream=True) for i in range(4)]
    assert all(e.side_streams == -1 for _, e in pairs)

    def job_of(e):
        def job():
            for _ in range(5):
                e.train_step(xd, yd, 35.0, 1e-3)
            return e.gate_timeouts()
        return job

    touts = T.run_concurrent([job_of(e) for _, e in pairs], 4)
    torch.cuda.synchronize()
    assert touts == [0, 0, 0, 0]
    for i, (_, e) in enumerate(pairs):
        assert torch.equal(e.params, ref[i]), i


@pytest.mark.parametrize("b", [8, 64, 512])
def test_single_stream_context_is_bitwise_the_default_one(b):
    """eae_config.side_streams = -1 (every kernel on the caller's stream, in dependency order): the same losses, gradients and
    parameters, bit for bit, as the default context with its two side streams -- joint steps, a reconstruction-only step, and the
    gradient-only entry point."""
    import eae_amd
    from eae_amd.engine import engine_for
    x, y = gu.make_images(b, 4100 + b)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = []
    for ss in ((0, -1, 1) if b == 64 else (0, -1)):          # default (two side streams), none, and -- at one size -- exactly one
        torch.manual_seed(321)
        m = eae_amd.SupervisedAutoencoder(64).cuda().train()
        if ss:
            m._eae_side_streams = ss
        e = engine_for(m, max_batch=b)
        assert e.side_streams == ss
        losses = []
        for step in range(3):
            e.train_step(xd, yd, 35.0, 2e-3)
            losses.append(e.loss_last.cpu().numpy().copy())
        e.train_step(xd, yd, 1.0, 2e-3, head=False)
        losses.append(e.loss_last.cpu().numpy().copy())
        e.grad_step(xd, yd, 35.0)
        torch.cuda.synchronize()
        assert e.gate_timeouts() == 0
        out.append((np.stack(losses), e.grads.cpu().numpy().copy(), e.params.cpu().numpy().copy(),
                    {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}))
    a = out[0]
    for c in out[1:]:
        assert np.array_equal(a[0], c[0]), (a[0], c[0])
        assert np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
        for k in a[3]:
            assert np.array_equal(a[3][k], c[3][k]), k
    assert np.isfinite(a[0]).all() and np.abs(a[1]).max() > 0


def test_grouped_grid_is_bitwise_the_sequential_one_with_the_groups_geometries(tmp_path):
    """grid_search_autoencoder(grouped=4): four configurations step in lockstep through ONE sequence of grouped launches per batch
    (include/eae.h eae_group_train_step / eae_group_forward).  Under torch.manual_seed the default model construction draws the same
    parameters in grid order as the sequential grid, and with the sequential run's launchers set to the group's tile geometries
    (eae_set_geometry_mult(4)) every configuration's curves, early-stopping epoch and the saved winner are bitwise equal.
    patience=1: members drop out of the group at different epochs, the others go on with unchanged arithmetic."""
    from eae_amd import _lib
    from eae_amd import train as T
    lib = _lib.load()
    tr, va = _loaders()
    kw = dict(alpha_values=(20, 35), lr_values=(1e-3, 2e-2), num_epochs=4, patience=1, verbose=True)
    seq_log, grp_log = [], []
    torch.manual_seed(4242)
    _lib.check(lib.eae_set_geometry_mult(4))
    try:
        seq = T.grid_search_autoencoder(tr, va, out_dir=str(tmp_path / "seq"), log=seq_log.append, **kw)
    finally:
        _lib.check(lib.eae_set_geometry_mult(1))
    torch.manual_seed(4242)
    grp = T.grid_search_autoencoder(tr, va, out_dir=str(tmp_path / "grp"), grouped=4, log=grp_log.append, **kw)
    assert seq["results"] == grp["results"], (seq["results"], grp["results"])
    assert (seq["best_alpha"], seq["best_lr"]) == (grp["best_alpha"], grp["best_lr"])
    assert seq["best_train_curve"] == grp["best_train_curve"] and seq["best_val_curve"] == grp["best_val_curve"]
    assert seq_log == grp_log
    assert all(np.isfinite(v) for v in seq["results"].values())
    sa, sb = torch.load(seq["best_path"]), torch.load(grp["best_path"])
    assert all(torch.equal(sa[k], sb[k]) for k in sa)


def test_two_concurrent_groups_are_bitwise_the_sequential_grid(tmp_path):
    """grid_search_autoencoder(grouped=4, concurrent_groups=2): two groups of four step at the same time from two host threads, every
    context with ONE side stream (2 x 2 streams = the four hardware queues).  A configuration's results do not depend on what runs
    beside it, nor on the number of side streams: curves, winner and saved weights are bitwise those of the sequential grid run
    with the groups' tile geometries."""
    from eae_amd import _lib
    from eae_amd import train as T
    lib = _lib.load()
    tr, va = _loaders()
    kw = dict(alpha_values=(20, 25, 30, 35), lr_values=(1e-3, 5e-3), num_epochs=2, patience=15, verbose=True)
    seq_log, par_log = [], []
    torch.manual_seed(777)
    _lib.check(lib.eae_set_geometry_mult(4))
    try:
        seq = T.grid_search_autoencoder(tr, va, out_dir=str(tmp_path / "seq"), log=seq_log.append, **kw)
    finally:
        _lib.check(lib.eae_set_geometry_mult(1))
    torch.manual_seed(777)
    par = T.grid_search_autoencoder(tr, va, out_dir=str(tmp_path / "par"), grouped=4, concurrent_groups=2, log=par_log.append, **kw)
    assert seq["results"] == par["results"], (seq["results"], par["results"])
    assert (seq["best_alpha"], seq["best_lr"]) == (par["best_alpha"], par["best_lr"]) and seq_log == par_log
    sa, sb = torch.load(seq["best_path"]), torch.load(par["best_path"])
    assert all(torch.equal(sa[k], sb[k]) for k in sa)


def test_hardware_queue_check_and_worker_streams():
    """eae_streams_share_queue: a stream shares its queue with itself; run_concurrent's worker streams (<= 4 workers) sit on pairwise
    different hardware queues and are the same objects in the next call with the same number of workers; reserving / releasing a
    stream is accepted.  (ROCm multiplexes a process's streams onto 4 hardware queues: DESIGN.md section 6.)"""
    from eae_amd import _lib
    from eae_amd import train as T
    lib = _lib.load()
    s = torch.cuda.Stream()
    assert lib.eae_streams_share_queue(s.cuda_stream, s.cuda_stream) == 1
    assert lib.eae_reserve_stream(s.cuda_stream, 1) == 0 and lib.eae_reserve_stream(s.cuda_stream, 0) == 0
    seen = []

    def job():
        seen.append(torch.cuda.current_stream().cuda_stream)
        return len(seen)
    T.run_concurrent([job, job, job], 3, static=True)
    first = sorted(seen)
    assert len(set(first)) == 3
    for i in range(3):
        for j in range(i + 1, 3):
            assert lib.eae_streams_share_queue(first[i], first[j]) == 0, (i, j)
    seen.clear()
    T.run_concurrent([job, job, job], 3, static=True)
    assert sorted(seen) == first
