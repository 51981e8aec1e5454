"""GPU: the flow of the reference's examples/run_classification_criteo.py (BASELINE config 1) through the ``deepctr`` facade —
same imports, same calls: CSV -> LabelEncoder / MinMaxScaler -> SparseFeat / DenseFeat -> DeepFM(...).compile / fit / predict —
on the committed 200-row data file.  With the weights of the reference-code fixture the prediction must equal the fixture's
output; fit() must run with the example's arguments and reduce the loss."""
import os

import numpy as np
import pytest

from tests.util import GOLDEN, assert_close, load_golden


def criteo_example_inputs():
    import pandas as pd
    from sklearn.preprocessing import LabelEncoder, MinMaxScaler

    from deepctr.feature_column import DenseFeat, SparseFeat, get_feature_names
    data = pd.read_csv(os.path.join(GOLDEN, "criteo_sample.txt"))
    sparse_features = ["C" + str(i) for i in range(1, 27)]
    dense_features = ["I" + str(i) for i in range(1, 14)]
    data[sparse_features] = data[sparse_features].fillna("-1")
    data[dense_features] = data[dense_features].fillna(0)
    for feat in sparse_features:
        data[feat] = LabelEncoder().fit_transform(data[feat])
    data[dense_features] = MinMaxScaler(feature_range=(0, 1)).fit_transform(data[dense_features])
    cols = [SparseFeat(feat, vocabulary_size=data[feat].max() + 1, embedding_dim=4) for feat in sparse_features] + \
           [DenseFeat(feat, 1) for feat in dense_features]
    return data, cols, get_feature_names(cols + cols)


@pytest.mark.gpu
def test_criteo_example_flow_through_the_facade(device):
    from sklearn.metrics import log_loss, roc_auc_score
    from sklearn.model_selection import train_test_split

    from deepctr.models import DeepFM
    data, cols, feature_names = criteo_example_inputs()
    target = ["label"]
    # (1) the fixture's weights -> the fixture's predictions, fed as pandas Series like the example does
    g = load_golden("model_deepfm_criteo_sample")
    model = DeepFM(cols, cols, task="binary", device=device)
    model.set_weights_by_name({k[2:]: v for k, v in g.items() if k.startswith("w/")})
    y = model.predict({name: data[name] for name in feature_names}, batch_size=256)
    assert_close(y, g["y"], rtol=1e-4, atol=1e-6, what="criteo_sample through the facade")
    # (2) the example's training calls, verbatim arguments
    train, test = train_test_split(data, test_size=0.2, random_state=2020)
    train_model_input = {name: train[name] for name in feature_names}
    test_model_input = {name: test[name] for name in feature_names}
    model = DeepFM(cols, cols, task="binary", device=device)
    model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy"])
    history = model.fit(train_model_input, train[target].values, batch_size=256, epochs=10, verbose=2, validation_split=0.2)
    assert len(history.history["loss"]) == 10 and len(history.history["val_loss"]) == 10
    assert history.history["loss"][-1] < history.history["loss"][0]
    pred_ans = model.predict(test_model_input, batch_size=256)
    assert pred_ans.shape == (len(test), 1) and np.isfinite(pred_ans).all()
    ll = log_loss(test[target].values, pred_ans.astype(np.float64), labels=[0, 1])
    auc = roc_auc_score(test[target].values, pred_ans)
    assert 0.0 < ll < 2.0 and 0.0 <= auc <= 1.0
