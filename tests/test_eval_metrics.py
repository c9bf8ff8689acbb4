"""evaluation reductions (callbacks.py:36-55).  CPU: the numpy oracle against a
fixture written through scikit-learn.  GPU: the HIP reductions against the fixture and against the oracle."""
import numpy as np
import pytest
import torch

from oracle import eval_metrics as E
from tests.util import golden


def test_oracle_matches_sklearn_fixture():
    g = golden("eval_metrics.npz")
    assert np.allclose(E.f1_samples(g["probs"], g["labels"], g["thresholds"]), g["f1"], rtol=0, atol=1e-15)
    a, w, c = E.average_precision(g["probs"], g["labels"])
    assert abs(a - float(g["ap_samples"])) < 1e-14 and abs(w - float(g["ap_weighted"])) < 1e-14
    assert np.allclose(c, g["ap_class"], rtol=0, atol=1e-14)
    assert c[14] == 0.0                                           # class without positives


@pytest.mark.gpu
def test_hip_metrics_match_sklearn_fixture(device):
    from dvt_amd import ops
    g = golden("eval_metrics.npz")
    p, l = torch.from_numpy(g["probs"]).cuda(), torch.from_numpy(g["labels"]).cuda()
    f1 = ops.f1_samples(p, l, g["thresholds"]).cpu().numpy()
    assert np.abs(f1 - g["f1"]).max() < 1e-6
    a, w, c = ops.average_precision(p, l)
    assert abs(float(a) - float(g["ap_samples"])) < 1e-6 and abs(float(w) - float(g["ap_weighted"])) < 1e-6
    assert np.abs(c.cpu().numpy() - g["ap_class"]).max() < 1e-6


@pytest.mark.gpu
def test_hip_metrics_large_and_callback(device):
    """20k validation samples with tied scores vs the oracle; the callback logs the reference's keys and resets."""
    from dvt_amd import ops
    from dvt_amd.metrics import TransformerEval
    rng = np.random.default_rng(3)
    N, C = 20000, 19
    y = (rng.random((N, C)) < 0.15).astype(np.uint8)
    s = np.round(rng.random((N, C)), 3).astype(np.float32)          # 1000 distinct values: many ties
    a, w, c = ops.average_precision(torch.from_numpy(s).cuda(), torch.from_numpy(y).cuda())
    ra, rw, rc = E.average_precision(s[:2000], y[:2000])
    a2, w2, c2 = ops.average_precision(torch.from_numpy(s[:2000]).cuda(), torch.from_numpy(y[:2000]).cuda())
    assert abs(float(a2) - ra) < 1e-6 and abs(float(w2) - rw) < 1e-6 and np.abs(c2.cpu().numpy() - rc).max() < 1e-6
    cls_full = np.array([E.average_precision_1d(s[:, k], y[:, k]) for k in (0, 7, 18)])
    assert np.abs(c.cpu().numpy()[[0, 7, 18]] - cls_full).max() < 1e-6

    class Mod:
        def __init__(self):
            self.running_logits = [torch.from_numpy(s[:64]).cuda(), torch.from_numpy(s[64:100]).cuda()]
            self.running_labels = [torch.from_numpy(y[:64]).cuda().int(), torch.from_numpy(y[64:100]).cuda().int()]
            self.logged = {}

        def log(self, k, v, **kw):
            self.logged[k] = v
    m = Mod()
    out = TransformerEval().on_validation_epoch_end(None, m)
    assert m.running_logits == [] and m.running_labels == []
    assert set(out) == set(m.logged) and "sklearn apr" in out and "val/online/f1@0.3" in out
    ref_f1 = E.f1_samples(s[:100], y[:100], [0.3])[0]
    assert abs(out["val/online/f1@0.3"] - ref_f1) < 1e-6
    assert abs(out["sklearn apr"] - E.average_precision(s[:100], y[:100])[0]) < 1e-6
