"""Shared wiring for the in-scope model constructors."""
from .. import ops
from ..engine import Model
from ..feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
from ..initializers import Zeros
from ..inputs import create_embedding_matrix
from ..layers.utils import Linear


def linear_columns(feature_columns):
    """get_linear_logit (reference feature_column.py:171-181): the same columns with embedding_dim=1 and
    Zeros() initialiser — i.e. a second, 1-wide set of tables named ``linear0sparse_emb_<name>``."""
    out = []
    for fc in feature_columns:
        if isinstance(fc, SparseFeat):
            out.append(fc._replace(embedding_dim=1, embeddings_initializer=Zeros()))
        elif isinstance(fc, VarLenSparseFeat):
            out.append(fc._replace(sparsefeat=fc.sparsefeat._replace(embedding_dim=1, embeddings_initializer=Zeros())))
        else:
            out.append(fc)
    return out


class FeatureModel(Model):
    """linear part + embedding stage + DNN + head: the skeleton DeepFM / DCN / xDeepFM share."""

    def build_linear(self, linear_feature_columns, seed):
        lin_cols = linear_columns(linear_feature_columns)
        self.linear_tables = create_embedding_matrix(lin_cols, 0, seed, prefix="linear0", device=self.device)
        for t in self.linear_tables.values():
            self._add(t)
        n_sparse = sum(1 for fc in lin_cols if not isinstance(fc, DenseFeat))
        n_dense = sum(fc.dimension for fc in lin_cols if isinstance(fc, DenseFeat))
        self.linear = None
        if n_dense > 0:
            self.linear = Linear(mode=2 if n_sparse > 0 else 1, seed=seed, device=self.device).build_for(n_dense)
            self._add(self.linear)
        return lin_cols

    def build_embeddings(self, dnn_feature_columns, seed, prefix=""):
        self.tables = create_embedding_matrix(dnn_feature_columns, 0, seed, prefix=prefix, device=self.device)
        for t in self.tables.values():
            self._add(t)

    def _stage_inputs(self, feed, staged):
        self.stage_plan.stage(feed, staged)

    def _pipeline(self, x, batch_size):
        """Large host feeds of fixed-length features: stage in chunks overlapped with the copies and the scoring."""
        from .. import _C, engine
        if type(self)._stage_inputs is not FeatureModel._stage_inputs or not batch_size:
            return None
        _C.require_device()
        feed = self._as_feed(x)
        n = self._num_rows(feed)
        plan = self.stage_plan.pipeline_plan(feed, n)
        if plan is None:
            return None
        bs = int(batch_size)
        chunk = bs * max(1, -(-engine._PIPELINE_CHUNK_ROWS // bs))     # whole batches per chunk
        staged = engine.Staged(n)
        return staged, self.stage_plan.stage_chunks(plan, staged, chunk), bs

    def _begin(self):
        self.stage_plan.refresh(self.linear.w('linear_kernel') if self.linear is not None else None)
        dnn = getattr(self, "dnn", None)
        if dnn is not None and getattr(dnn, "bn_layers", None):
            dnn.bn_params()             # BatchNormalization scale / shift follow the current weights (in place)

    def _check_status(self):
        ops.check_status(self.stage_plan.status(), "embedding lookup in model %s" % self.name)

    def _logits_to_add(self, ws):
        add = []
        if self.stage_plan.has_linear:
            add.append(ws["lin"])
        if "lin2" in ws:
            add.append(ws["lin2"])
        return add
