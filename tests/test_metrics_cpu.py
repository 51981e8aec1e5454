"""CPU: the metrics behind Model.evaluate / val_<metric> of fit() against scikit-learn's definitions (the reference's examples score
predictions with sklearn.metrics.log_loss / roc_auc_score: examples/run_classification_criteo.py:50-52)."""
import numpy as np
import pytest

from deepctr_amd.engine import Model


def test_metrics_match_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    rng = np.random.RandomState(3)
    y = (rng.rand(500) > 0.6).astype(np.float64)
    p = np.clip(rng.rand(500) * 0.7 + 0.3 * y * rng.rand(500), 0, 1)
    p[::7] = np.round(p[::7], 1)                                   # ties
    assert abs(Model._metric("auc", p, y) - sk.roc_auc_score(y, p)) < 1e-12
    assert abs(Model._metric("binary_crossentropy", p, y) - sk.log_loss(y, np.clip(p, 1e-7, 1 - 1e-7))) < 1e-12
    assert abs(Model._metric("mse", p, y) - sk.mean_squared_error(y, p)) < 1e-15
    assert abs(Model._metric("mae", p, y) - sk.mean_absolute_error(y, p)) < 1e-15
    assert abs(Model._metric("acc", p, y) - sk.accuracy_score(y > 0.5, p > 0.5)) < 1e-15
    assert np.isnan(Model._metric("auc", p, np.ones_like(y)))
    with pytest.raises(NotImplementedError):
        Model._metric("cosine_similarity", p, y)
