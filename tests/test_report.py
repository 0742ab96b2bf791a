"""Reporting helpers against sklearn -- the library the reference calls at R.md:3188, 3216 -- used here only
as the checker -- and against the one table the reference's report holds (R.md:3218-3234)."""
import json

import numpy as np
import pytest

from eae_amd import report as R


def _case(seed, n=3000, c=10):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, c, n)
    flip = rng.random(n) < 0.3
    p = np.where(flip, rng.integers(0, c, n), y)
    return y, p


def test_confusion_matrix_and_report_match_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    for seed in (0, 1):
        y, p = _case(seed)
        np.testing.assert_array_equal(R.confusion_matrix(y, p), sk.confusion_matrix(y, p))
        assert R.classification_report(y, p, digits=4) == sk.classification_report(y, p, digits=4)
    # a class that is never predicted (zero_division branch) and a class only predicted
    y = np.array([0, 0, 1, 1, 2, 2, 2]); p = np.array([0, 1, 1, 1, 0, 0, 4])
    np.testing.assert_array_equal(R.confusion_matrix(y, p), sk.confusion_matrix(y, p))
    m = R.class_metrics(y, p)
    pr, rc, f1, sup = sk.precision_recall_fscore_support(y, p, zero_division=0)
    np.testing.assert_allclose(m["precision"], pr); np.testing.assert_allclose(m["recall"], rc)
    np.testing.assert_allclose(m["f1"], f1); np.testing.assert_array_equal(m["support"], sup)


def test_report_reproduces_reference_table():
    """R.md:3218-3234 prints precision/recall/support per class; a confusion matrix with those row sums and diagonal
    (off-diagonal mass placed so the column sums give the printed precisions) must reproduce the printed summary lines."""
    support = np.array([321, 293, 295, 302, 295, 279, 270, 315, 314, 316])
    recall = np.array([0.9346, 0.0375, 0.6610, 0.9040, 0.9525, 0.4731, 0.7111, 0.9365, 0.8057, 0.9810])
    tp = np.rint(recall * support).astype(int)
    assert tp.sum() == 2242                                  # accuracy 0.7473 * 3000 (R.md:3231)
    y = np.repeat(np.arange(10), support)
    p = y.copy()
    # put every misclassified sample into class 9 / class 0 (the two attractors named at R.md:3249-3252)
    off = 0
    for c in range(10):
        wrong = support[c] - tp[c]
        p[off + tp[c]: off + support[c]] = 9 if c != 9 else 0
        off += support[c]
    m = R.class_metrics(y, p)
    assert abs(m["accuracy"] - 0.7473) < 5e-5
    np.testing.assert_allclose(m["recall"], recall, atol=5e-5)
    assert abs(m["macro"][1] - 0.7397) < 5e-5                # macro recall R.md:3232
    assert abs(m["weighted"][1] - 0.7473) < 5e-5             # weighted recall == accuracy R.md:3233
    txt = R.classification_report(y, p, digits=4)
    assert "accuracy                         0.7473      3000" in txt


def test_heatmap_and_best_config(tmp_path):
    alphas, lrs = (20, 35), (0.001, 0.005)
    res = {f"alpha={a}, lr={lr}": float(a) / 100 + lr for a in alphas for lr in lrs}
    res["alpha=35, lr=0.005"] = 0.5397                        # the reference's best cell (R.md:2441)
    path = tmp_path / "validation_losses.json"
    path.write_text(json.dumps(res, indent=4))
    loaded = R.load_validation_losses(str(path))
    hm = R.loss_heatmap(loaded, alphas, lrs)
    assert hm.shape == (2, 2) and hm[1, 1] == 0.5397 and hm[0, 0] == pytest.approx(0.201)
    res2 = dict(res); res2["alpha=20, lr=0.001"] = 0.1
    assert R.best_config(res2) == (20.0, 0.001, 0.1)
