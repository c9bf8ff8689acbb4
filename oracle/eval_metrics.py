"""CPU restatement of the evaluation reductions.

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Reference call sites: src/callbacks/callbacks.py:36-55 -- on the concatenated ``running_logits`` (sigmoid
probabilities, frame_transformer.py:331-333) and ``running_labels``:
  * ``f1_score(labels, probs > t, average="samples", zero_division=0)`` for t in 0, 0.1 .. 0.8,
  * ``average_precision_score(labels, probs, average="samples")`` and ``average="weighted"``.
The arithmetic is scikit-learn's (third party, unpinned by the reference; 1.7.2 is installed here and the
restatement is pinned against it by tests/golden/eval_metrics.npz and by direct comparison in the tests):
  samples-F1:  per row 2|P & T| / (|P| + |T|), 0 when both are empty; mean over rows.
  AP:          scores sorted descending, one operating point per distinct score value;
               AP = sum_k (R_k - R_{k-1}) P_k with P_k = tp_k / k-th prefix length, R_k = tp_k / positives;
               no positives -> 0.  "samples": mean of the row APs; "weighted": class APs weighted by support.
"""
from __future__ import annotations

import numpy as np


def f1_samples(probs: np.ndarray, labels: np.ndarray, thresholds) -> np.ndarray:
    lab = labels.astype(bool)
    out = []
    for t in thresholds:
        pred = probs > np.float32(t)
        tp = (pred & lab).sum(1).astype(np.float64)
        den = pred.sum(1) + lab.sum(1)
        f = np.where(den > 0, 2.0 * tp / np.maximum(den, 1), 0.0)
        out.append(f.mean())
    return np.asarray(out)


def average_precision_1d(scores: np.ndarray, labels: np.ndarray) -> float:
    order = np.argsort(-scores, kind="mergesort")
    s, l = scores[order], labels[order].astype(np.int64)
    total = int(l.sum())
    if total == 0:
        return 0.0
    ap, tp, r_prev = 0.0, 0, 0.0
    n = len(s)
    for i in range(n):
        tp += int(l[i])
        if i == n - 1 or s[i + 1] != s[i]:
            r = tp / total
            ap += (r - r_prev) * (tp / (i + 1))
            r_prev = r
    return ap


def average_precision(probs: np.ndarray, labels: np.ndarray):
    """-> (samples average, support-weighted class average, per-class AP)."""
    N, C = probs.shape
    rows = np.array([average_precision_1d(probs[i], labels[i]) for i in range(N)])
    cls = np.array([average_precision_1d(probs[:, c], labels[:, c]) for c in range(C)])
    support = labels.astype(np.int64).sum(0)
    weighted = float((cls * support).sum() / support.sum()) if support.sum() > 0 else 0.0
    return float(rows.mean()), weighted, cls
