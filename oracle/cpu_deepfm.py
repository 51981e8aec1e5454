"""ORACLE — TEST INFRASTRUCTURE ONLY (bench.py's ``cpu_baseline`` leg and tests may import this).

CPU restatement of the reference's DeepFM forward as the op SEQUENCE Keras executes it
(deepctr/models/deepfm.py:42-65): one Embedding gather per feature and per table set (inputs.py:101-117;
feature_column.py:171-210 for the 1-wide linear tables), Concat, FM's two reductions
(layers/interaction.py:588-604), Linear (layers/utils.py:160-175), three Dense+ReLU (layers/core.py:189-208),
Dense(1), Add, PredictionLayer (layers/core.py:250-259) — on torch-CPU ops with all host cores, because
TensorFlow itself cannot be installed in this environment.  ``kind: "port"`` in bench.py.
Checked against oracle/ref_models.py in tests/test_cpu_baseline.py.
"""
import torch


class CpuDeepFM(object):
    def __init__(self, weights, n_sparse, n_dense, names=None):
        names = names or ["C%d" % i for i in range(1, n_sparse + 1)]
        t = lambda k: torch.from_numpy(weights[k]).float()  # noqa: E731
        self.tables = [t("sparse_emb_%s/embeddings" % n) for n in names]
        self.lin_tables = [t("linear0sparse_emb_%s/embeddings" % n) for n in names]
        self.lin_kernel = t("linear/linear_kernel") if n_dense else None
        self.kernels, self.biases = [], []
        i = 0
        while "dnn/kernel%d" % i in weights:
            self.kernels.append(t("dnn/kernel%d" % i))
            self.biases.append(t("dnn/bias%d" % i))
            i += 1
        self.head = t("dense/kernel")
        self.global_bias = t("prediction_layer/global_bias")

    @torch.no_grad()
    def forward(self, ids, dense):
        """ids: list of F int64 [B] tensors; dense: list of [B,1] float tensors."""
        embs = [torch.index_select(w, 0, i).unsqueeze(1) for w, i in zip(self.tables, ids)]        # F x [B,1,E]
        lins = [torch.index_select(w, 0, i).unsqueeze(1) for w, i in zip(self.lin_tables, ids)]    # F x [B,1,1]
        sparse_lin = torch.cat(lins, dim=-1)                                                       # [B,1,F]
        dense_in = torch.cat(dense, dim=-1) if dense else None
        linear_logit = sparse_lin.sum(dim=-1)                                                      # [B,1]
        if dense_in is not None:
            linear_logit = linear_logit + dense_in @ self.lin_kernel
        x = torch.cat(embs, dim=1)                                                                 # [B,F,E]
        square_of_sum = x.sum(dim=1, keepdim=True).pow(2)
        sum_of_square = (x * x).sum(dim=1, keepdim=True)
        fm_logit = 0.5 * (square_of_sum - sum_of_square).sum(dim=2)                                # [B,1]
        h = x.flatten(1)
        if dense_in is not None:
            h = torch.cat([h, dense_in], dim=-1)
        for w, b in zip(self.kernels, self.biases):
            h = torch.relu(h @ w + b)
        logit = linear_logit + h @ self.head + fm_logit
        return torch.sigmoid(logit + self.global_bias).reshape(-1, 1)
