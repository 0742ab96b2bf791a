"""Reporting helpers of the reference workflow: the validation-loss heat-map table of the alpha x lr grid
(R.md:2415-2425), the loss curves (R.md:2460-2469), the confusion matrix (R.md:3188-3190) and the per-class report the
notebook prints with sklearn's `classification_report(..., digits=4)` (R.md:3216-3237).

Numbers only, computed with NumPy (sklearn is not needed to get them).  Drawing the figures is out of scope (SURVEY.md section 2:
plotting).
"""
import json

import numpy as np


def load_validation_losses(path="models_best/validation_losses.json"):
    """The JSON written by `grid_search_autoencoder` / the reference (R.md:718-729): {"alpha=A, lr=LR": best_val_loss}."""
    with open(path, "r") as f:
        return json.load(f)


def loss_heatmap(results, alpha_values, lr_values):
    """heatmap[i, j] = best validation loss of (alpha_values[i], lr_values[j])   (R.md:2419-2425)."""
    hm = np.zeros((len(alpha_values), len(lr_values)))
    for i, a in enumerate(alpha_values):
        for j, lr in enumerate(lr_values):
            hm[i, j] = results[f"alpha={a}, lr={lr}"]
    return hm


def best_config(results):
    """(alpha, lr, loss) of the lowest validation loss; first entry wins ties, as in the reference's strict `<` (R.md:701)."""
    best = None
    for key, v in results.items():
        if best is None or v < best[2]:
            a, lr = key.split(", ")
            best = (float(a.split("=")[1]), float(lr.split("=")[1]), v)
    return best


def confusion_matrix(labels, preds, num_classes=None):
    """cm[i, j] = number of samples of true class i predicted as j (sklearn.metrics.confusion_matrix semantics for integer
    classes; when num_classes is None the classes are the sorted union of labels and predictions)."""
    labels = np.asarray(labels).astype(np.int64).ravel()
    preds = np.asarray(preds).astype(np.int64).ravel()
    if labels.shape != preds.shape:
        raise ValueError("labels and preds must have the same length")
    if num_classes is None:
        classes = np.unique(np.concatenate([labels, preds]))
        idx = {c: i for i, c in enumerate(classes.tolist())}
        li = np.array([idx[c] for c in labels.tolist()], dtype=np.int64)
        pi = np.array([idx[c] for c in preds.tolist()], dtype=np.int64)
        n = len(classes)
    else:
        li, pi, n = labels, preds, int(num_classes)
    cm = np.zeros((n, n), dtype=np.int64)
    np.add.at(cm, (li, pi), 1)
    return cm


def class_metrics(labels, preds):
    """Per-class precision / recall / f1 / support plus accuracy, macro and weighted averages (zero_division -> 0)."""
    labels = np.asarray(labels).astype(np.int64).ravel()
    preds = np.asarray(preds).astype(np.int64).ravel()
    classes = np.unique(np.concatenate([labels, preds]))
    cm = confusion_matrix(labels, preds)
    tp = np.diag(cm).astype(np.float64)
    pred_n = cm.sum(0).astype(np.float64)
    true_n = cm.sum(1).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        prec = np.where(pred_n > 0, tp / pred_n, 0.0)
        rec = np.where(true_n > 0, tp / true_n, 0.0)
        f1 = np.where(prec + rec > 0, 2 * prec * rec / (prec + rec), 0.0)
    total = true_n.sum()
    w = true_n / max(total, 1.0)
    return {
        "classes": classes, "precision": prec, "recall": rec, "f1": f1, "support": true_n.astype(np.int64),
        "accuracy": float(tp.sum() / max(total, 1.0)),
        "macro": (float(prec.mean()), float(rec.mean()), float(f1.mean())),
        "weighted": (float((prec * w).sum()), float((rec * w).sum()), float((f1 * w).sum())),
        "total": int(total),
    }


def classification_report(labels, preds, digits=4):
    """Text table in the layout sklearn prints for the notebook's call (R.md:3216-3237)."""
    m = class_metrics(labels, preds)
    names = [str(c) for c in m["classes"].tolist()]
    width = max(max(len(n) for n in names), len("weighted avg"), digits)
    head = "{:>{w}s} ".format("", w=width) + "".join(" {:>9}".format(h) for h in ("precision", "recall", "f1-score", "support"))
    lines = [head, ""]
    row = "{:>{w}s} " + " {:>9.{d}f}" * 3 + " {:>9}"
    for i, n in enumerate(names):
        lines.append(row.format(n, m["precision"][i], m["recall"][i], m["f1"][i], int(m["support"][i]), w=width, d=digits))
    lines.append("")
    lines.append("{:>{w}s} ".format("accuracy", w=width) + " {:>9} {:>9}".format("", "") + " {:>9.{d}f} {:>9}".format(m["accuracy"], m["total"], d=digits))
    lines.append(row.format("macro avg", *m["macro"], m["total"], w=width, d=digits))
    lines.append(row.format("weighted avg", *m["weighted"], m["total"], w=width, d=digits))
    return "\n".join(lines) + "\n"
