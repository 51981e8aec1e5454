"""The reference's examples/run_classification_criteo.py flow (feature columns -> model -> compile -> fit -> predict) on
synthetic Criteo-shaped data, on one MI355X.  Needs the built HIP library and a GPU (there is no CPU path).

    python -m deepctr_amd.build && python examples/run_classification_synthetic.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepctr_amd.feature_column import DenseFeat, SparseFeat, get_feature_names  # noqa: E402
from deepctr_amd.models import DeepFM  # noqa: E402


def main(n=200000, vocab=1000, emb=16):
    rng = np.random.RandomState(2020)
    sparse_features = ["C%d" % i for i in range(1, 27)]
    dense_features = ["I%d" % i for i in range(1, 14)]
    data = {name: rng.randint(0, vocab, n).astype(np.int32) for name in sparse_features}
    data.update({name: rng.rand(n).astype(np.float32) for name in dense_features})          # MinMax-scaled in the reference
    label = ((data["C1"] % 2) ^ (data["I1"] > 0.5)).astype(np.float32)                        # something learnable

    fixlen_feature_columns = [SparseFeat(feat, vocabulary_size=vocab, embedding_dim=emb) for feat in sparse_features] + \
                             [DenseFeat(feat, 1) for feat in dense_features]
    dnn_feature_columns = fixlen_feature_columns
    linear_feature_columns = fixlen_feature_columns
    feature_names = get_feature_names(linear_feature_columns + dnn_feature_columns)

    split = int(0.8 * n)
    train_input = {name: data[name][:split] for name in feature_names}
    test_input = {name: data[name][split:] for name in feature_names}

    model = DeepFM(linear_feature_columns, dnn_feature_columns, task='binary')
    model.compile("adam", "binary_crossentropy", metrics=['binary_crossentropy'])
    history = model.fit(train_input, label[:split], batch_size=4096, epochs=3, verbose=1, validation_split=0.1)
    pred = model.predict(test_input, batch_size=4096)
    eps = 1e-7
    p = np.clip(pred.reshape(-1), eps, 1 - eps)
    y = label[split:]
    print("test LogLoss %.4f" % float(-(y * np.log(p) + (1 - y) * np.log(1 - p)).mean()))
    print("loss per epoch:", ["%.4f" % v for v in history.history["loss"]])


if __name__ == "__main__":
    main()
