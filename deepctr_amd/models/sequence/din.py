"""DIN — same signature as ``deepctr.models.sequence.din.DIN`` (reference deepctr/models/sequence/din.py:20-96).

Per batch: pooled non-history sequences (``dctr_embed_pool``) + fused gather of every SparseFeat into the DNN
input; query / key lookups (``dctr_embed_lookup``, keys keep their per-position mask_zero mask and the masks of
all history features are AND-ed as keras ``Concat.compute_mask`` does, reference layers/utils.py:198-228);
``dctr_din_attn_pool_fwd`` (LocalActivationUnit MLP on f32 MFMA + masked weighted sum) writes the attention
output straight into its slot of the DNN input; ``dctr_mlp_fwd`` finishes (DNN + Dense(1) + sigmoid).
DIN has no linear term."""
import torch

from ... import ops
from ...engine import EmbeddingStage, prehashed_on_host
from ...feature_column import SparseFeat, VarLenSparseFeat
from ...layers.base import name_scope
from ...layers.core import DNN, Dense, PredictionLayer
from ...layers.sequence import AttentionSequencePoolingLayer
from .._common import FeatureModel


class _DIN(FeatureModel):
    def __init__(self, dnn_feature_columns, history_feature_list, dnn_use_bn, dnn_hidden_units, dnn_activation,
                 att_hidden_size, att_activation, att_weight_normalization, dnn_dropout, seed, task, device):
        super(_DIN, self).__init__("DIN", list(dnn_feature_columns), device, task)
        self.history_feature_list = list(history_feature_list)
        hist_names = ["hist_" + n for n in self.history_feature_list]
        sparse = [fc for fc in dnn_feature_columns if isinstance(fc, SparseFeat)]
        varlen = [fc for fc in dnn_feature_columns if isinstance(fc, VarLenSparseFeat)]
        self.history_cols = [fc for fc in varlen if fc.name in hist_names]
        self.query_cols = [fc for fc in sparse if fc.name in self.history_feature_list]
        if not self.history_cols or not self.query_cols:
            raise ValueError("DIN needs history_feature_list features and their 'hist_<name>' sequence columns")
        self.key_dim = sum(fc.embedding_dim for fc in self.history_cols)
        self.query_dim = sum(fc.embedding_dim for fc in self.query_cols)
        if self.key_dim != self.query_dim:
            raise ValueError("query width %d != key width %d" % (self.query_dim, self.key_dim))
        T = set(fc.maxlen for fc in self.history_cols)
        if len(T) != 1:
            raise ValueError("history sequences must share one maxlen")
        self.T = T.pop()
        with name_scope():
            self.linear_tables, self.linear = {}, None
            self.build_embeddings(dnn_feature_columns, seed)
            # DNN input = [all SparseFeat embeddings, pooled non-history sequences, attention output, dense]
            # (din.py:70-89); sparse ids of history features are hashed with mask_zero=True (din.py:70-71)
            self.stage_plan = EmbeddingStage(self.tables, {}, [], dnn_feature_columns,
                                             mask_feat_list=tuple(self.history_feature_list),
                                             extra_dims=(("hist", self.key_dim),),
                                             skip_varlen=tuple(hist_names), device=self.device)
            self.attention = AttentionSequencePoolingLayer(att_hidden_size, att_activation,
                                                           weight_normalization=att_weight_normalization,
                                                           supports_masking=True, device=self.device)
            self.attention.build_for(self.key_dim)
            self._add(self.attention)
            self.dnn = self._add(DNN(dnn_hidden_units, dnn_activation, 0, dnn_dropout, dnn_use_bn, seed=seed,
                                     device=self.device).build_for(self.stage_plan.in_dim))
            last = dnn_hidden_units[-1] if len(dnn_hidden_units) else self.stage_plan.in_dim
            self.dense = self._add(Dense(1, use_bias=False, seed=seed, device=self.device).build_for(last))
            self.prediction = self._add(PredictionLayer(task, device=self.device).build_for())
        self._buf = {}
        # query / key lookups folded into the attention kernels (dctr_din_attn_gather_fwd: the [B, T, E] keys never reach HBM);
        # False, or a shape outside those kernels: dctr_embed_lookup_multi + dctr_din_attn_pool_fwd
        self.fold_lookups = True
        self._fold_failed = False

    def _fold_lookups_ok(self):
        if not self.fold_lookups or self._fold_failed or len(self.history_cols) > 2 or len(self.history_cols) != len(self.query_cols):
            return False
        cols = list(self.history_cols) + list(self.query_cols)
        if any(fc.use_hash and not prehashed_on_host(fc) for fc in cols):
            return False
        dims = set(fc.embedding_dim for fc in cols)
        return len(dims) == 1 and dims.pop() % 16 == 0 and not self.attention.return_score

    def _attention_folded(self, staged, lo, hi, ws, out):
        """AttentionSequencePoolingLayer over rows [lo, hi) with the lookups inside the kernels; False when the library declines."""
        q_ids = [staged.ids[row, lo:hi] for row in self._query_rows]
        h_ids = [staged.seq[fc.name][lo:hi] for fc in self.history_cols]
        h_tab = [self.tables[fc.embedding_name].embeddings for fc in self.history_cols]
        q_tab = [self.tables[fc.embedding_name].embeddings for fc in self.query_cols]
        mz = [bool(self.tables[fc.embedding_name].mask_zero) for fc in self.history_cols]
        la = self.attention.local_att
        r = ops.din_attention_gather(h_ids, q_ids, h_tab, q_tab, mz, la.dnn.kernels, la.dnn.biases, la.w("kernel"), la.w("bias"),
                                     self.attention.att_activation, la.dnn.dice_params(),
                                     weight_normalization=self.attention.weight_normalization, out=out,
                                     out_stride=self.stage_plan.out_stride, status=ws["status"],
                                     compact=self.attention.compact_positions)
        if r is None:
            self._fold_failed = True
            return False
        return True

    def _stage_inputs(self, feed, staged):
        self.stage_plan.stage(feed, staged)
        for fc in self.history_cols:
            self.stage_plan.stage_varlen(feed, staged, fc)
        # query ids: rows of the id matrix that belong to the history features
        self._query_rows = []
        for fc in self.query_cols:
            for i, f in enumerate(self.stage_plan.fields):
                if f.kind == "sparse" and f.fc.name == fc.name:
                    self._query_rows.append(i)
                    break

    def _forward(self, staged, lo, hi, out):
        sp = self.stage_plan
        ws = sp.run(staged, lo, hi)
        hist_off = sp.extra_offsets["hist"]
        if not (self._fold_lookups_ok() and self._attention_folded(staged, lo, hi, ws, ws["dnn_in"][:, hist_off:])):
            bufs = self._attention_inputs(staged, lo, hi, ws)
            self.attention.run(bufs["q"], bufs["k"], bufs["m"], out=ws["dnn_in"][:, hist_off:], out_stride=sp.out_stride)
        ops.mlp(ws["dnn_in"], self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                head_w=self.dense.w('kernel'), global_bias=self.prediction.w('global_bias'),
                sigmoid_out=self.task == "binary", in_dim=sp.in_dim, out=out)

    def _attention_inputs(self, staged, lo, hi, ws):
        """Query [B,E'] / key [B,T,E'] embeddings and the key mask [B,T] of rows [lo, hi) (din.py:62-76); also records the
        lookups of the history features (``bufs['key_lookups']``: (feature, ids, hash_mode, first key column)) for the
        training step's scatter."""
        B = hi - lo
        bufs = self._buf.get(B)
        if bufs is None:
            if len(self._buf) >= 4:            # ragged remainder sizes (N % span) must not pile up per-B buffers
                self._buf.clear()
            bufs = self._buf[B] = dict(
                q=torch.zeros(B, self.query_dim, dtype=torch.float32, device=self.device),
                k=torch.zeros(B, self.T, self.key_dim, dtype=torch.float32, device=self.device),
                m=torch.ones(B, self.T, dtype=torch.uint8, device=self.device))
        st = ws["status"]
        # query features and behaviour sequences: ONE launch (dctr_embed_lookup_multi); the first mask_zero sequence's
        # lookup writes the attention mask = conjunction of all mask_zero sequences' (id != 0)
        lookups, col = [], 0
        for fc, row in zip(self.query_cols, self._query_rows):
            table = self.tables[fc.embedding_name].embeddings
            hm = 2 if (fc.use_hash and not prehashed_on_host(fc)) else 0
            lookups.append(dict(idx=staged.ids[row, lo:hi], table=table, hash_mode=hm, out=bufs["q"][:, col:]))
            col += fc.embedding_dim
        col = 0
        masked = [fc for fc in self.history_cols if self.tables[fc.embedding_name].mask_zero]
        bufs["key_lookups"] = []
        for fc in self.history_cols:
            emb = self.tables[fc.embedding_name]
            hm = 2 if (fc.use_hash and not prehashed_on_host(fc)) else 0
            lk = dict(idx=staged.seq[fc.name][lo:hi], table=emb.embeddings, hash_mode=hm, out=bufs["k"][:, :, col:])
            bufs["key_lookups"].append((fc, lk["idx"], hm, col))
            if masked and fc is masked[0]:
                lk["mask"] = bufs["m"]
            lookups.append(lk)
            col += fc.embedding_dim
        if len(masked) > 5:
            raise NotImplementedError("DIN with more than five mask_zero behaviour sequences is outside the fused lookup's limits")
        extra = [staged.seq[fc.name][lo:hi] for fc in masked[1:]]
        for c0 in range(0, len(lookups), 8):            # eight lookups per launch
            ops.embed_lookup_multi(lookups[c0:c0 + 8], extra_mask_ids=extra, status=st)
        return bufs                         # bufs["m"]: all ones when no history feature masks zero (never written then)


def DIN(dnn_feature_columns, history_feature_list, dnn_use_bn=False, dnn_hidden_units=(256, 128, 64),
        dnn_activation='relu', att_hidden_size=(80, 40), att_activation="dice", att_weight_normalization=False,
        l2_reg_dnn=0, l2_reg_embedding=1e-6, dnn_dropout=0, seed=1024, task='binary', device=None):
    """Instantiates the Deep Interest Network architecture on the MI355X forward path."""
    m = _DIN(dnn_feature_columns, history_feature_list, dnn_use_bn, dnn_hidden_units, dnn_activation,
             att_hidden_size, att_activation, att_weight_normalization, dnn_dropout, seed, task, device)
    # l2 regularisers of the reference constructor (din.py:56-57, :91); the attention unit has none (sequence.py:243-245)
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": 0.0, "dnn": float(l2_reg_dnn)}
    return m
