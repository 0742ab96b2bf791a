"""Bookkeeping of the fit loops (sample-weighted epoch means are the stepper's job on the device; here: early stopping,
best tracking, grid JSON schema, the reference's aliasing quirk) with scripted steppers -- no GPU needed.
Expected values are computed by hand from R.md:656-711 and R.md:2639-2704."""
import json
import os

import torch

import eae_amd
from eae_amd import train as T


class ScriptedAE:
    device = None

    def __init__(self, train_losses, val_losses):
        self.t, self.v, self.i, self.phase = train_losses, val_losses, 0, None
        self.calls = []

    def begin(self):
        self.n = 0

    def train_step(self, x, y):
        self.phase = "t"; self.n += len(x); self.calls.append(("t", len(x)))

    def eval_step(self, x, y):
        self.phase = "v"; self.n += len(x); self.calls.append(("v", len(x)))

    def end(self):
        e = self.i // 2
        out = (self.t[e] if self.i % 2 == 0 else self.v[e]), self.n
        self.i += 1
        return out


def _loader(sizes):
    return [(torch.zeros(b, 1), torch.zeros(b, dtype=torch.int64)) for b in sizes]


def test_early_stopping_strict_improvement():
    # val: improves at epochs 1,2; equal value at epoch 3 is NOT an improvement (strict <, R.md:690); patience 3
    val = [5.0, 4.0, 4.0, 4.5, 4.2, 3.0, 1.0]
    st = ScriptedAE([9, 8, 7, 6, 5, 4, 3], val)
    logs = []
    r = T.fit_autoencoder(_loader([64, 48]), _loader([64, 56]), alpha=35, lr=1e-3, num_epochs=7, patience=3, stepper=st,
                          model=None, log=logs.append)
    assert r["epochs"] == 5                      # epochs 3,4,5 fail to improve -> stop after the 5th
    assert r["best_val_loss"] == 4.0
    assert r["val_curve"] == val[:5]
    assert logs[-1] == "Early stopping triggered."
    assert logs[0] == "[AE α=35 LR=0.001] Epoch 1 | TrainLoss=9.0000 | ValLoss=5.0000"      # print format R.md:686-687
    # every batch of both loaders visited once per epoch, train before val
    assert st.calls[:4] == [("t", 64), ("t", 48), ("v", 64), ("v", 56)]


def test_grid_search_bookkeeping(tmp_path):
    table = {(20, 0.1): 3.0, (20, 0.2): 2.5, (30, 0.1): 2.5, (30, 0.2): 2.7}

    def fake_fit(tr, va, alpha, lr, **kw):
        return {"model": None, "train_curve": [1.0], "val_curve": [table[(alpha, lr)]], "best_val_loss": table[(alpha, lr)], "epochs": 1}

    out = T.grid_search_autoencoder([], [], alpha_values=(20, 30), lr_values=(0.1, 0.2), out_dir=str(tmp_path), verbose=False, fit_fn=fake_fit)
    assert (out["best_alpha"], out["best_lr"]) == (20, 0.2)          # first strict minimum wins (R.md:702)
    js = json.load(open(os.path.join(tmp_path, "validation_losses.json")))
    assert js == {"alpha=20, lr=0.1": 3.0, "alpha=20, lr=0.2": 2.5, "alpha=30, lr=0.1": 2.5, "alpha=30, lr=0.2": 2.7}


def test_grid_search_concurrent_same_bookkeeping_and_log_order(tmp_path):
    """concurrent=K runs the configurations on K host threads (each on its own stream on a GPU): the results table, the winner
    (first strict minimum in GRID order, R.md:702) and the log (a configuration's lines together, in grid order) are those of the
    sequential loop, whatever order the threads finish in."""
    import threading
    import time
    table = {(20, 0.1): 3.0, (20, 0.2): 2.5, (30, 0.1): 2.5, (30, 0.2): 2.7}
    seen = []

    def fake_fit(tr, va, alpha, lr, log=print, **kw):
        time.sleep(0.05 if alpha == 20 else 0.0)              # the first row finishes LAST
        seen.append(threading.get_ident())
        log(f"[AE α={alpha} LR={lr}] Epoch 1 | TrainLoss=1.0000 | ValLoss={table[(alpha, lr)]:.4f}")
        return {"model": None, "train_curve": [1.0], "val_curve": [table[(alpha, lr)]], "best_val_loss": table[(alpha, lr)], "epochs": 1}

    logs = {}
    outs = {}
    for k in (1, 4):
        seen.clear()
        logs[k] = []
        d = tmp_path / f"k{k}"
        outs[k] = T.grid_search_autoencoder([], [], alpha_values=(20, 30), lr_values=(0.1, 0.2), out_dir=str(d), verbose=True, fit_fn=fake_fit,
                                            log=logs[k].append, concurrent=k, device="cpu")
        if k > 1:
            assert len(set(seen)) > 1                            # really ran on several threads
    assert logs[1] == logs[4]
    assert (outs[4]["best_alpha"], outs[4]["best_lr"]) == (outs[1]["best_alpha"], outs[1]["best_lr"]) == (20, 0.2)
    assert outs[4]["results"] == outs[1]["results"]
    assert json.load(open(tmp_path / "k4" / "validation_losses.json")) == json.load(open(tmp_path / "k1" / "validation_losses.json"))


class ScriptedGroup:
    """K scripted members behind the group-stepper interface (begin / train_step / eval_step / end with the active index list)."""
    device = None

    def __init__(self, members):
        self.m = members
        self.calls = []

    def begin(self, active):
        for k in active:
            self.m[k].begin()

    def train_step(self, x, y, active):
        self.calls.append(("t", len(x), tuple(active)))
        for k in active:
            self.m[k].train_step(x, y)

    def eval_step(self, x, y, active):
        self.calls.append(("v", len(x), tuple(active)))
        for k in active:
            self.m[k].eval_step(x, y)

    def end(self, active):
        return [self.m[k].end() for k in active]


def test_group_fit_is_the_sequential_fit_per_member_and_drops_stopped_members():
    """fit_autoencoder_group keeps every member's curves / best / early stopping exactly as fit_autoencoder does for it alone
    (R.md:686-697), walks the loaders once per epoch for the whole group, and stops stepping a member once it has stopped."""
    vals = [[5.0, 4.0, 4.0, 4.5, 4.2, 3.0, 1.0], [3.0, 3.5, 3.6, 3.7, 1.0, 1.0, 1.0], [9.0, 8.0, 7.0, 6.0, 5.0, 4.0, 3.0]]
    trains = [[9, 8, 7, 6, 5, 4, 3], [1, 1, 1, 1, 1, 1, 1], [2, 2, 2, 2, 2, 2, 2]]
    cfgs = [(35, 1e-3), (20, 2e-3), (40, 5e-3)]
    alone, alone_logs = [], []
    for k in range(3):
        lg = []
        alone.append(T.fit_autoencoder(_loader([64, 48]), _loader([64, 56]), alpha=cfgs[k][0], lr=cfgs[k][1], num_epochs=7, patience=3,
                                       stepper=ScriptedAE(trains[k], vals[k]), model=None, log=lg.append))
        alone_logs.append(lg)
    st = ScriptedGroup([ScriptedAE(trains[k], vals[k]) for k in range(3)])
    lines = [[] for _ in range(3)]
    rs = T.fit_autoencoder_group(_loader([64, 48]), _loader([64, 56]), cfgs, num_epochs=7, patience=3, stepper=st, models=None, logs=lines)
    for k in range(3):
        for key in ("train_curve", "val_curve", "best_val_loss", "epochs"):
            assert rs[k][key] == alone[k][key], (k, key)
        assert lines[k] == alone_logs[k]
    assert [r["epochs"] for r in rs] == [5, 4, 7]
    # epoch 1: all three members on every batch; epoch 5: member 1 (stopped after 4) is no longer stepped; epoch 6: member 0 gone too
    assert st.calls[:4] == [("t", 64, (0, 1, 2)), ("t", 48, (0, 1, 2)), ("v", 64, (0, 1, 2)), ("v", 56, (0, 1, 2))]
    assert st.calls[16] == ("t", 64, (0, 2)) and st.calls[20] == ("t", 64, (2,))


def test_grid_search_grouped_same_bookkeeping_and_log_order(tmp_path):
    """grouped=K: the grid is cut into groups of K in grid order, results / JSON / winner / log order are those of the sequential grid."""
    table = {(20, 0.1): 3.0, (20, 0.2): 2.5, (30, 0.1): 2.5, (30, 0.2): 2.7, (40, 0.1): 2.4, (40, 0.2): 2.4}
    seen = []

    def fake_fit(tr, va, alpha, lr, log=print, **kw):
        log(f"fit {alpha} {lr}")
        return {"model": None, "train_curve": [1.0], "val_curve": [table[(alpha, lr)]], "best_val_loss": table[(alpha, lr)], "epochs": 1}

    def fake_group_fit(tr, va, cfgs, logs=None, **kw):
        seen.append(list(cfgs))
        out = []
        for k, (a, lr) in enumerate(cfgs):
            logs[k].append(f"fit {a} {lr}")
            out.append({"model": None, "train_curve": [1.0], "val_curve": [table[(a, lr)]], "best_val_loss": table[(a, lr)], "epochs": 1})
        return out

    seq_logs, grp_logs = [], []
    a = T.grid_search_autoencoder(None, None, alpha_values=(20, 30, 40), lr_values=(0.1, 0.2), out_dir=str(tmp_path / "seq"), fit_fn=fake_fit,
                                  log=seq_logs.append)
    b = T.grid_search_autoencoder(None, None, alpha_values=(20, 30, 40), lr_values=(0.1, 0.2), out_dir=str(tmp_path / "grp"), grouped=4,
                                  group_fit_fn=fake_group_fit, log=grp_logs.append)
    assert seen == [[(20, 0.1), (20, 0.2), (30, 0.1), (30, 0.2)], [(40, 0.1), (40, 0.2)]]
    assert a["results"] == b["results"] and (a["best_alpha"], a["best_lr"]) == (b["best_alpha"], b["best_lr"]) == (40, 0.1)
    assert seq_logs == grp_logs
    assert json.load(open(tmp_path / "seq" / "validation_losses.json")) == json.load(open(tmp_path / "grp" / "validation_losses.json"))
    # two groups at a time from two host threads: same bookkeeping, same log order
    seen.clear()
    par_logs = []
    c = T.grid_search_autoencoder(None, None, alpha_values=(20, 30, 40), lr_values=(0.1, 0.2), out_dir=str(tmp_path / "par"), grouped=2,
                                  concurrent_groups=2, group_fit_fn=fake_group_fit, log=par_logs.append, device="cpu")
    assert sorted(seen) == [[(20, 0.1), (20, 0.2)], [(30, 0.1), (30, 0.2)], [(40, 0.1), (40, 0.2)]]
    assert a["results"] == c["results"] and (c["best_alpha"], c["best_lr"]) == (40, 0.1) and par_logs == seq_logs


class ScriptedMLP:
    device = None

    def __init__(self, clf, val_acc):
        self.clf, self.val_acc, self.k, self.epoch = clf, val_acc, 0, 0

    def begin(self):
        pass

    def train_step(self, x, y):
        with torch.no_grad():
            self.clf.net[7].bias += 1.0           # "training" mutates the live weights

    def eval_step(self, x, y):
        pass

    def end(self):
        self.k += 1
        phase = (self.k - 1) % 2                   # 0 = train, 1 = val (the final test call comes last)
        e = (self.k - 1) // 2
        if phase == 1 and e < len(self.val_acc):
            return 0.5, self.val_acc[e], 10
        return 0.5, 0.25, 10


def test_mlp_best_state_aliasing_quirk():
    val = [0.3, 0.6, 0.5]
    for alias in (True, False):
        clf = eae_amd.MLP(64)
        with torch.no_grad():
            clf.net[7].bias.zero_()
        st = ScriptedMLP(clf, val)
        r = T.fit_mlp(_loader([4]), _loader([4]), _loader([4]), lr=1e-3, num_epochs=3, clf=clf, stepper=st, alias_best=alias,
                      verbose=False)
        assert r["best_val_acc"] == 0.6 and r["val_acc"] == val
        # one train batch per epoch -> bias = epoch count. Best epoch = 2.
        got = float(clf.net[7].bias[0])
        if alias:      # reference behaviour (R.md:2683): shallow copy aliases the live tensors -> final-epoch weights
            assert got == 3.0
        else:
            assert got == 2.0


def test_evaluate_and_extract_contract_on_cpu_raises():
    # no CPU fallback: calling the product path without a HIP device must fail loudly
    import pytest
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    clf = eae_amd.MLP(64)
    with pytest.raises(RuntimeError):
        eae_amd.evaluate(clf, _loader([4]))
    m = eae_amd.SupervisedAutoencoder(64)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m(torch.zeros(2, 3, 64, 64))
