"""HIP training step — SURVEY.md §8(f) rank 1: backward + optimizer of the hot path on HIP kernels.

Covers the DeepFM family (embedding gather [+ linear] [+ FM] -> DNN -> Dense(1) -> PredictionLayer: DeepFM, WDL, FNN;
NFM and PNN add their interaction layer's forward / backward kernel in front of the DNN; AFM has its
AFMLayer instead of the DNN), DCN, DCNMix (``dctr_crossnet_mix_bwd``), xDeepFM and DIN (attention input / weighted sum /
lookup scatter kernels; Dice as tf.keras runs it under fit(): ``dctr_dice_train_fwd`` + the BatchNormalization backward inside
``dctr_mlp_bwd``), sequence features included: forward = ``dctr_embed_pool`` per sequence feature + ``dctr_embed_gather_fm`` +
``dctr_mlp_fwd`` (activations saved), then ``dctr_bce_grad`` → ``dctr_mlp_bwd`` → ``dctr_embed_gather_fm_bwd`` →
``dctr_embed_pool_bwd`` → ``dctr_opt_multi`` (one launch over every parameter).  No torch autograd, no torch
optimizer: PyTorch only owns the buffers.  ``dnn_dropout``, ``dnn_use_bn``, a Dice DNN, softmax-normalised DIN attention and
several FM groups run on this step as well (layer-by-layer forms of the DNN); ``afm_dropout`` and a PReLU DNN keep the
torch-autograd step of ``training.py``.

Semantics follow tf.keras as the reference uses it (``model.compile("adam", "binary_crossentropy")``,
examples/run_classification_criteo.py:44-50; also "adagrad", "rmsprop", "sgd" by name with tf.keras' defaults): Adam lr
1e-3, beta 0.9 / 0.999, epsilon 1e-7, NON-lazy on embeddings
(every row of a table decays every step, as ``_resource_apply_sparse`` does), L2 regularisers of the constructor
(``l2_reg_embedding``, ``l2_reg_linear``, ``l2_reg_dnn``) added to the gradients.
"""
import math

import torch

from . import ops


from .training import BN_MOMENTUM  # noqa: E402  (keras BatchNormalization default momentum, as the torch step uses)


def supported(model):
    """Can this model train on the HIP step?  (DeepFM family, fixed-length features, relu/linear/sigmoid/tanh DNN.)"""
    sp = getattr(model, "stage_plan", None)
    dnn = getattr(model, "dnn", None)
    kind = type(model).__name__
    if sp is None or kind not in ("_DeepFM", "_DCN", "_DCNMix", "_xDeepFM", "_NFM", "_PNN", "_AFM", "_DIN"):
        return False
    if kind == "_DIN":
        # attention unit: Dice (moving statistics, as the torch step uses) or sigmoid / relu; plain weighted sum only
        la = model.attention.local_att
        if la.dnn.activation not in ("dice", "Dice", "sigmoid", "relu", "tanh", "linear"):
            return False
        if not la.dnn.kernels or getattr(la.dnn, "dropout_rate", 0) or getattr(la.dnn, "use_bn", False):
            return False
        # Dice runs as tf.keras runs it under fit(): BatchNormalization in training mode — this batch's statistics, gradients
        # through them, stored statistics moved (dctr_dice_train_fwd + dctr_mlp_bwd's dice_batch_*);
        # model.hip_dice_stored_statistics = True keeps the stored statistics instead (tests of the inference-form backward)
    # (round 6: features only the linear part sees — linear_feature_columns is its own list in every constructor — train on the HIP step:
    #  their first-order weights receive d logit through the second gather's backward, HipTrainer.step)
    # (round 6: DIN over any key width — the attention unit trains on the materialised [B * T, 4 E'] input, dctr_embed_lookup_bwd scatters
    #  any width (its sorted-tile form up to 64 columns, plain atomics past that); tables whose width is not a multiple of 4 carry no
    #  touched-group marks)
    # (round 6: any embedding width — widths that are not a multiple of 4 or exceed 64, embedding_dim="auto" — the scatter kernels walk a
    # row in chunks, element per lane where rows are not 16-B aligned; such tables carry no touched-group marks: dense optimizer pass)
    if kind == "_xDeepFM" and getattr(model, "cin", None) is not None and not ops.cin_supported(
            len(sp.fields), model.cin_dim, list(model.cin.layer_size), model.cin.split_half, model.cin.activation):
        return False                        # (the library's answer — dctr_cin_fwd_supported; round 6: any embedding width, in slices of d past 128)
    # (round 6: a matrix CrossNet of any width — past the [16, dim] LDS tiles of the one-kernel training forward, ~832 columns, i.e. Criteo
    #  at embedding_dim 32, the forward runs layer by layer on dctr_sgemm + dctr_crossnet_matrix_step straight into saved_u / saved_x:
    #  _cross_fwd; dctr_crossnet_bwd's matrix form has no width of its own)
    if (kind == "_DCN" and getattr(getattr(model, "cross", None), "parameterization", None) == "vector" and sp.in_dim > 8192
            and int(getattr(model.cross, "layer_num", 0)) > 8):
        return False                        # (rows past the register file — 8192 columns — run the closed form of the vector recurrence, up to 8
                                            #  layers; the backward takes any width: layer by layer past 2048 columns / 48 L d bytes of LDS)
    if len(sp.fm_group_names) > 1:          # further FM groups (DeepFM / AFM fm_group): their logits ride on the head's four `add` slots
        n_add = int(bool(sp.has_linear)) + len(sp.fm_group_names)
        if kind not in ("_DeepFM", "_AFM") or n_add > 4:
            return False
    if kind == "_AFM":                      # no DNN: linear logit + AFMLayer per group (or the gather's FM group)
        return not any(getattr(layer, "dropout_rate", 0) for layer in model.afm_layers)
    if sp.extra_offsets and kind not in ("_NFM", "_PNN", "_DIN"):   # the interaction columns those reserve in dnn_in
        return False
    if kind in ("_DeepFM", "_xDeepFM", "_NFM", "_PNN", "_DIN") and (dnn is None or not dnn.kernels):
        return False
    if kind == "_xDeepFM" and model.cin is not None and model.cin.activation not in ("relu", "linear", "sigmoid", "tanh"):
        return False
    if dnn is not None:
        # dnn_dropout > 0 / dnn_use_bn=True: the DNN runs layer by layer (dctr_dnn_train_layer_fwd / _bwd behind each dense part)
        # dnn_activation="dice" (layers/activation.py:37-64 under training=True): layer by layer as well — dctr_dice_train_fwd behind
        # each dense part, dctr_mlp_bwd's Dice form with the batch statistics and the saved pre-activations on the way back
        if dnn.activation in ("dice", "Dice"):
            if getattr(dnn, "dropout_rate", 0) or getattr(dnn, "bn_layers", None) or not dnn.kernels:
                return False
            if getattr(dnn, "output_activation", None) not in (None, dnn.activation) or dnn.dice_params() is None:
                return False
        elif dnn.activation not in ("relu", "linear", "sigmoid", "tanh") or not dnn.kernels:
            return False
        if getattr(dnn, "output_activation", None) not in (None, dnn.activation):
            return False
    return True


class _Param(object):
    __slots__ = ("w", "m", "v", "g", "l2", "touched")

    def __init__(self, w, l2=0.0):
        self.w = w
        self.m = torch.zeros_like(w)
        self.v = torch.zeros_like(w)
        self.g = torch.zeros_like(w)
        self.l2 = float(l2)
        self.touched = None     # embedding tables: one byte per 16-B group of g (include/dctr.h, dctr_field_grad_t)

    def track_rows(self):
        """Embedding table [vocab, dim % 4 == 0]: the scatter kernels mark the 16-B groups of ``g`` they add to, and the optimizer
        step reads / clears only those (a 4096-row batch touches a few percent of a 1e5-row table)."""
        if self.touched is None and self.w.dim() == 2 and self.w.shape[1] % 4 == 0 and self.w.is_contiguous():
            self.touched = torch.zeros(self.w.numel() // 4, dtype=torch.uint8, device=self.w.device)
        return self


class _Frozen(object):
    """A weight that takes no part in training (Embedding.trainable == False): ``g`` is None = a NULL gradient table."""
    __slots__ = ("w", "g", "l2", "touched")

    def __init__(self, w):
        self.w, self.g, self.l2, self.touched = w, None, 0.0, None

    def track_rows(self):
        return self


# tf.keras defaults of the optimizers model.compile() takes by name (optimizer_v2/*.py)
OPT_DEFAULTS = {"adam": dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7), "adagrad": dict(lr=1e-3, eps=1e-7, init_acc=0.1),
                "rmsprop": dict(lr=1e-3, beta2=0.9, eps=1e-7), "sgd": dict(lr=1e-2)}


class HipTrainer(object):
    def __init__(self, model, optimizer="adam", lr=None, beta1=None, beta2=None, eps=None):
        if not supported(model):
            raise ValueError("model is outside the HIP training step's family (see training_hip.supported)")
        optimizer = optimizer.lower()
        if optimizer not in OPT_DEFAULTS:
            raise ValueError("optimizer %r not supported (adam, adagrad, rmsprop, sgd)" % optimizer)
        d = OPT_DEFAULTS[optimizer]
        self.kind = optimizer
        self.model = model
        self.lr = float(d["lr"] if lr is None else lr)
        self.b1 = float(d.get("beta1", 0.0) if beta1 is None else beta1)
        self.b2 = float(d.get("beta2", 0.0) if beta2 is None else beta2)
        self.eps = float(d.get("eps", 0.0) if eps is None else eps)
        self.init_acc = float(d.get("init_acc", 0.0))
        self.t = 0
        sp = model.stage_plan
        reg = getattr(model, "regularizers", {})
        l2e, l2l, l2d = reg.get("embedding", 0.0), reg.get("linear", 0.0), reg.get("dnn", 0.0)
        self.params = []
        by_ptr = {}
        # SparseFeat(trainable=False) (reference inputs.py:25: emb.trainable = feat.trainable; FAQ "pretrained embeddings"):
        # a frozen table gets NO gradient buffer (the backward kernels skip NULL gradient tables), no optimizer segment and
        # no l2 decay — fit() leaves it bit-identical
        frozen = set()
        for embs in (getattr(model, "tables", None) or {}, getattr(model, "linear_tables", None) or {}):
            for emb in embs.values():
                if not getattr(emb, "trainable", True):
                    frozen.add(emb.embeddings.data_ptr())

        def param(t, l2=0.0):
            key = t.data_ptr()
            if key not in by_ptr:
                if key in frozen:
                    by_ptr[key] = _Frozen(t)
                else:
                    by_ptr[key] = _Param(t, l2)
                    self.params.append(by_ptr[key])
            return by_ptr[key]

        # per field: (gradient table, gradient of the linear table).  Sequence features are pooled by dctr_embed_pool into
        # per-batch buffers the gather reads as identity fields: their gradients land in per-batch buffers too
        # (_buffers) and dctr_embed_pool_bwd scatters them on to the tables.
        self.field_params = []
        for f in sp.fields:
            pt = param(f.table, l2e).track_rows()
            pl = param(f.lin_table, l2l) if f.lin_table is not None else None
            self.field_params.append((f, pt, pl))
        # features of the linear part alone (feature_column.py:171-210 builds the linear logit from linear_feature_columns only): their
        # 1-wide tables are parameters too; the second gather of the forward (EmbeddingStage.run_lin_only) has its own backward in step()
        from .feature_column import VarLenSparseFeat as _VarLen
        self.lin_only_params = [(fc, param(model.linear_tables[fc.embedding_name].embeddings, l2l)) for fc in sp.lin_only]
        self.lin_only_varlen = [(fc, p) for fc, p in self.lin_only_params if isinstance(fc, _VarLen)]
        # Linear.kernel (dense features of the linear part): the forward reads a copy permuted into dense-matrix column
        # order (EmbeddingStage.refresh); the backward kernel scatters straight into the real kernel's gradient
        self.p_dense_lin, self.dense_rows = None, None
        if model.linear is not None and sp.n_dense and sp.n_lin_dense and sp.has_linear:
            self.p_dense_lin = param(model.linear.w("linear_kernel"), l2l)
            self.dense_rows = torch.as_tensor(sp.dense_lin_rows, dtype=torch.int32, device=model.device)
        self.is_mix = type(model).__name__ == "_DCNMix"
        self.is_dcn = type(model).__name__ in ("_DCN", "_DCNMix")
        self.is_afm = type(model).__name__ == "_AFM"
        self.p_afm = []
        if self.is_afm:
            for layer in model.afm_layers:      # l2_reg_att applies to attention_W only (interaction.py:100)
                self.p_afm.append((param(layer.w("attention_W"), getattr(layer, "l2_reg_w", 0.0)), param(layer.w("attention_b")),
                                   param(layer.w("projection_h")), param(layer.w("projection_p"))))
        self.is_din = type(model).__name__ == "_DIN"
        self.p_att = None
        if self.is_din:
            la = model.attention.local_att
            dice = la.dnn.dice_params()
            # no regulariser on the attention unit (AttentionSequencePoolingLayer builds it with l2_reg=0, sequence.py:243-245)
            self.p_att = dict(kernels=[param(k) for k in la.dnn.kernels], biases=[param(b) for b in la.dnn.biases],
                              alphas=[param(d[0]) for d in dice] if dice else None, out_w=param(la.w("kernel")),
                              out_b=param(la.w("bias")))
            # history tables receive the key gradients through dctr_embed_lookup_bwd
            self.p_hist = [param(model.tables[fc.embedding_name].embeddings, l2e).track_rows() for fc in model.history_cols]
            # columns of the query embeddings inside the DNN input (dq is added there; the gather backward scatters it)
            qcol = []
            for fc in model.query_cols:
                f = next(f for f in sp.fields if f.kind == "sparse" and f.fc.name == fc.name)
                qcol.extend(range(f.out_offset, f.out_offset + f.dim))
            self.qcol = torch.as_tensor(qcol, dtype=torch.int32, device=model.device)
        self.is_nfm = type(model).__name__ == "_NFM"
        self.is_pnn = type(model).__name__ == "_PNN"
        self.p_kernels = [param(k, l2d) for k in model.dnn.kernels] if model.dnn is not None else []
        self.p_biases = [param(b) for b in model.dnn.biases] if model.dnn is not None else []
        self.p_head = param(model.dense.w("kernel")) if getattr(model, "dense", None) is not None else None
        # DNN(dropout_rate > 0 / use_bn=True): training-mode BatchNormalization (gamma / beta trained, stored statistics moved by the
        # forward) and Dropout (reference layers/core.py:196-208) — the layer-by-layer form of _dnn_forward / _dnn_backward
        dnn = model.dnn
        self.drop_rate = float(getattr(dnn, "dropout_rate", 0) or 0) if dnn is not None else 0.0
        self.bn_layers = list(getattr(dnn, "bn_layers", None) or []) if dnn is not None else []
        self.slow_dnn = bool(self.drop_rate > 0 or self.bn_layers)
        self.dice_dnn = dnn is not None and dnn.activation in ("dice", "Dice")
        self.p_dice_alpha = [param(d[0]) for d in dnn.dice_params()] if self.dice_dnn else []
        self.p_bn = [(param(b.w("gamma")) if b.scale else None, param(b.w("beta")) if b.center else None) for b in self.bn_layers]
        self.drop_base = int(getattr(dnn, "seed", 1024) or 0) * 0x9E3779B1 + 12345 if dnn is not None else 0
        self.n_steps = 0            # forward passes so far: the dropout masks of a step are a function of (drop_base, n_steps, layer)
        self.is_xdeepfm = type(model).__name__ == "_xDeepFM"
        self.p_cin_f = self.p_cin_b = self.p_head1 = None
        if self.is_xdeepfm and model.cin is not None:
            self.p_cin_f = [param(f, reg.get("cin", 0.0)) for f in model.cin.filters]
            self.p_cin_b = [param(b) for b in model.cin.biases]
            self.p_head1 = param(model.dense_1.w("kernel"))
        self.p_cross_k = self.p_cross_b = None
        self.penalty_acc, self.penalty_rows = None, 0   # (fit(): device float64 accumulator of rows * l2 penalties; rows of the next update)
        self._cross_one_kernel = {}         # (rows, dim, layers) -> dctr_crossnet_fwd_supported's answer for the training forward
        self.p_mix = None
        if self.is_mix and model.cross is not None:
            # CrossNetMix: U / V / C stacked over layers, the experts' gating kernels, the biases — five packed parameter
            # tensors in the C ABI's layout; the layer's Keras-named weights become views of them.  l2 on U / V / C only
            # (the reference regularises U_list / V_list / C_list, interaction.py:481-500)
            U, V, C, G, Bb = model.cross.packed()
            l2c = reg.get("cross", 0.0)
            self.p_mix = [param(U.clone(), l2c), param(V.clone(), l2c), param(C.clone(), l2c), param(G.clone()), param(Bb.clone())]
            self.bind_cross_views()
        elif self.is_dcn and model.cross is not None:
            # CrossNet's per-layer kernels / biases become views of one packed tensor each (the layout the C ABI takes),
            # so that one parameter segment covers them and the layer keeps its Keras-named weights
            ks, bs = model.cross.packed()
            self.p_cross_k, self.p_cross_b = param(ks.clone(), reg.get("cross", 0.0)), param(bs.clone())
            self.bind_cross_views()
        self.p_gbias = param(model.prediction.w("global_bias")) if model.prediction.use_bias else None
        self._buf = {}
        if self.init_acc:
            for p in self.params:
                p.v.fill_(self.init_acc)            # Adagrad's initial_accumulator_value
        self.segs, self.n_segs, self.max_n = ops.make_adam_segments([(p.w, p.m, p.v, p.g, p.l2, p.touched) for p in self.params],
                                                                    model.device)

    def _buffers(self, B):
        b = self._buf.get(B)
        if b is None:
            dev = self.model.device
            sp = self.model.stage_plan
            units = [k.shape[1] for k in self.model.dnn.kernels] if self.model.dnn is not None else []
            if len(self._buf) >= 4:            # ragged remainder sizes (N % span) must not pile up per-B buffers
                self._buf.clear()
            b = self._buf[B] = {
                "dstack": torch.empty(B, (self.model.width + 3) // 4 * 4, dtype=torch.float32, device=dev) if self.is_dcn else None,
                "maps": torch.empty(B, self.model.cin_out_dim, dtype=torch.float32, device=dev) if self.p_cin_f else None,
                "dmaps": torch.empty(B, self.model.cin_out_dim, dtype=torch.float32, device=dev) if self.p_cin_f else None,
                "cin_logit": torch.empty(B, dtype=torch.float32, device=dev) if self.p_cin_f else None,
                # layer activations written by the forward CIN kernel for dctr_cin_bwd (else one recompute GEMM per layer)
                # (dctr_cin_fwd rejects save_y beyond the 2-GiB buffer-descriptor range: such steps recompute in dctr_cin_bwd)
                "cin_y": [torch.empty(B * self.model.cin_dim, h, dtype=torch.float32, device=dev)
                          for h in self.model.cin.layer_size]
                if (self.p_cin_f and B * self.model.cin_dim * max(self.model.cin.layer_size) * 4 < 2 ** 31) else None,
                # (None: dctr_cin_bwd re-runs the forward into its own workspace — whichever route the shape takes, round 6)
                "acts": [torch.empty(B, n, dtype=torch.float32, device=dev) for n in units],
                "pre": [torch.empty(B, n, dtype=torch.float32, device=dev) for n in units] if (self.slow_dnn or self.dice_dnn) else None,
                "dpre": [torch.empty(B, n, dtype=torch.float32, device=dev) for n in units] if self.slow_dnn else None,
                "bn_stat": [(torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev))
                            for n in units] if self.bn_layers else None,
                "pred": torch.empty(B, dtype=torch.float32, device=dev),
                "dlogit": torch.empty(B, dtype=torch.float32, device=dev),
                "dx": torch.empty(B, sp.out_stride, dtype=torch.float32, device=dev),
                "loss": torch.zeros(1, dtype=torch.float32, device=dev),
                "pooled_g": {}, "pooled_lin_g": {},
            }
            entries = []
            for f, pt, pl in self.field_params:
                if f.kind == "pooled":
                    b["pooled_g"][f.fc.name] = torch.zeros(B, f.dim, dtype=torch.float32, device=dev)
                    if pl is not None:
                        b["pooled_lin_g"][f.fc.name] = torch.zeros(B, dtype=torch.float32, device=dev)
                    entries.append((b["pooled_g"][f.fc.name], b["pooled_lin_g"].get(f.fc.name)))
                else:
                    entries.append((pt.g, None if pl is None else pl.g, pt.touched))
            b["field_grads"] = ops.make_field_grads(entries, dev)
            if self.lin_only_params:
                # the second gather's fields (EmbeddingStage: desc2): a SparseFeat scatters d logit into its linear table's gradient, a
                # pooled sequence into a per-batch vector that dctr_embed_pool_bwd carries on to the table
                b["lin2_pool_g"] = {fc.name: torch.zeros(B, 1, dtype=torch.float32, device=dev) for fc, _ in self.lin_only_varlen}
                entries2 = [(None, b["lin2_pool_g"][fc.name].reshape(-1) if fc.name in b["lin2_pool_g"] else p.g) for fc, p in self.lin_only_params]
                b["field_grads2"] = ops.make_field_grads(entries2, dev)
        return b

    def bind_cross_views(self):
        """(Re-)point CrossNet's Keras-named per-layer weights at views of the packed parameter tensors."""
        cr = self.model.cross
        d = cr.dim
        if self.p_mix is not None:
            U, V, C, G, Bb = (p.w for p in self.p_mix)
            for i in range(cr.layer_num):
                cr._weights['U_list%d' % i], cr._weights['V_list%d' % i], cr._weights['C_list%d' % i] = U[i], V[i], C[i]
                cr._weights['bias%d' % i] = Bb[i].view(d, 1)
            for e, dense in enumerate(cr.gating):
                dense._weights['kernel'] = G[e].view(d, 1)
            return
        for i in range(cr.layer_num):
            cr._weights['kernel%d' % i] = self.p_cross_k.w[i].view(d, -1)
            cr._weights['bias%d' % i] = self.p_cross_b.w[i].view(d, 1)

    # ---- the model's DNN (+ Dense(1) head) ------------------------------------------------------------------------------
    def dropout_seed(self, layer):
        """Seed of DNN layer ``layer``'s dropout mask in the CURRENT step (dctr_dnn_train_layer_t.dropout_seed)."""
        return (self.drop_base + self.n_steps * 1000003 + layer * 7919) & 0xFFFFFFFFFFFFFFFF

    def _bn_dict(self, l, buf):
        if not self.bn_layers:
            return None
        b = self.bn_layers[l]
        return dict(gamma=b.w("gamma") if b.scale else None, beta=b.w("beta") if b.center else None,
                    moving_mean=b.w("moving_mean"), moving_var=b.w("moving_variance"), eps=b.epsilon, momentum=b.momentum,
                    batch_mean=buf["bn_stat"][l][0], batch_var=buf["bn_stat"][l][1])

    def _dnn_forward(self, x, in_dim, buf, out, head=True, add=(), binary=False):
        """model.dnn over x[:, :in_dim] with the activations saved in buf["acts"]; ``head``: + Dense(1) + add + global bias (+ sigmoid)
        -> out [B]; headless: the last layer's activations -> out (a 2-D view)."""
        model, dnn = self.model, self.model.dnn
        gb = None if self.p_gbias is None else self.p_gbias.w
        if self.dice_dnn:
            # Dice under training=True needs the statistics of ALL rows of a layer before its activation: layer by layer
            dice = dnn.dice_params()
            stats, xin, kin = [], x, in_dim
            for l, (W, b) in enumerate(zip(dnn.kernels, dnn.biases)):
                ops.mlp(xin, [W], [b], "linear", in_dim=kin, out=buf["pre"][l])
                alpha, mmean, mvar = dice[l]
                stats.append(ops.dice_train_fwd(buf["pre"][l], alpha, mmean, mvar, buf["acts"][l], eps=1e-9, momentum=BN_MOMENTUM))
                xin, kin = buf["acts"][l], W.shape[1]
            buf["dice_batch"] = stats
            if head:
                ops.mlp(xin, [], [], "linear", head_w=self.p_head.w, add=list(add), global_bias=gb, sigmoid_out=binary, in_dim=kin, out=out)
            else:
                out[:, :kin].copy_(xin)             # (the stack's tail may carry alignment padding)
            return
        if not self.slow_dnn:
            if head:
                ops.mlp(x, dnn.kernels, dnn.biases, dnn.activation, head_w=self.p_head.w, add=list(add), global_bias=gb,
                        sigmoid_out=binary, in_dim=in_dim, out=out, save_acts=buf["acts"])
            else:
                ops.mlp(x, dnn.kernels, dnn.biases, dnn.activation, in_dim=in_dim, out=out, save_acts=buf["acts"])
            return
        # training-mode BatchNormalization / Dropout: dense part (one-layer linear launch) -> dctr_dnn_train_layer_fwd, layer by layer
        L = len(dnn.kernels)
        xin, kin = x, in_dim
        for l, (W, b) in enumerate(zip(dnn.kernels, dnn.biases)):
            ops.mlp(xin, [W], [b], "linear", in_dim=kin, out=buf["pre"][l])
            h = out if (not head and l == L - 1) else buf["acts"][l]
            ops.dnn_train_layer(buf["pre"][l], dnn.activation, h=h, bn=self._bn_dict(l, buf), dropout_rate=self.drop_rate,
                                dropout_seed=self.dropout_seed(l))
            xin, kin = h, W.shape[1]
        if head:
            ops.mlp(xin, [], [], "linear", head_w=self.p_head.w, add=list(add), global_bias=gb, sigmoid_out=binary, in_dim=kin, out=out)

    def _dnn_backward(self, x, in_dim, buf, dx, dlogit=None, d_out=None):
        """Backward of _dnn_forward: weight gradients accumulated, dx [B, >= in_dim] written.  With a head ``dlogit`` [B] comes in,
        headless ``d_out`` (2-D view) = gradient w.r.t. the last layer's activations."""
        model, dnn = self.model, self.model.dnn
        dk, db = [p.g for p in self.p_kernels], [p.g for p in self.p_biases]
        if self.dice_dnn:
            ops.mlp_bwd(x, in_dim, dnn.kernels, buf["acts"], "dice", self.p_head.w if dlogit is not None else None, dlogit, dk, db,
                        self.p_head.g if dlogit is not None else None, dx=dx, d_out=d_out, biases=dnn.biases, dice=dnn.dice_params(),
                        d_dice_alpha=[p.g for p in self.p_dice_alpha], dice_batch=buf["dice_batch"], saved_z=buf["pre"],
                        workspace=buf.setdefault("mlp_bwd_ws_dice", {}))
            return
        if not self.slow_dnn:
            # the weight-gradient launches go to a second stream (dctr_mlp_bwd_args_t.dw_stream): they run beside the embedding
            # scatter / CIN / CrossNet backward that follow on the main stream; step() joins before the optimizer
            side = self._side_stream()
            wsd = buf.setdefault("mlp_bwd_ws", {})
            if dlogit is not None:
                ops.mlp_bwd(x, in_dim, dnn.kernels, buf["acts"], dnn.activation, self.p_head.w, dlogit, dk, db, self.p_head.g, dx=dx,
                            dw_stream=side, workspace=wsd)
            else:
                ops.mlp_bwd(x, in_dim, dnn.kernels, buf["acts"], dnn.activation, None, None, dk, db, None, dx=dx, d_out=d_out,
                            dw_stream=side, workspace=wsd)
            self._side_used = side is not None
            return
        L = len(dnn.kernels)
        units = [k.shape[1] for k in dnn.kernels]
        if dlogit is not None:              # Dense(1): dH_last = dlogit (x) head_w, d_head_w += h_last^T dlogit
            dh = buf["dpre"][L - 1]
            ops.dense1_bwd(buf["acts"][L - 1], units[-1], self.p_head.w, dlogit, dh, self.p_head.g)
        else:
            dh = d_out
        for l in range(L - 1, -1, -1):
            dz = buf["dpre"][l]
            pg = self.p_bn[l] if self.p_bn else (None, None)
            ops.dnn_train_layer(buf["pre"][l], dnn.activation, bn=self._bn_dict(l, buf), dropout_rate=self.drop_rate,
                                dropout_seed=self.dropout_seed(l), dh=dh, dz=dz,
                                d_gamma=None if pg[0] is None else pg[0].g, d_beta=None if pg[1] is None else pg[1].g)
            xin, kin = (x, in_dim) if l == 0 else (buf["acts"][l - 1], units[l - 1])
            dst = dx if l == 0 else buf["dpre"][l - 1]
            # dense part of the layer: d_bias += colsum(dz), dW += x^T dz, dH_{l-1} = dz W^T  (one-layer headless linear dctr_mlp_bwd)
            # (one cached workspace per layer: the call would otherwise allocate one per layer and step)
            ops.mlp_bwd(xin, kin, [dnn.kernels[l]], [buf["acts"][l]], "linear", None, None, [dk[l]], [db[l]], None, dx=dst, d_out=dz,
                        workspace=buf.setdefault("mlp_bwd_ws_l%d" % l, {}))
            dh = dst

    side_stream = True          # False: everything on the caller's stream (A/B in scripts/bench_train.py --no-side-stream)

    def _side_stream(self):
        if not self.side_stream or self.model.device.type != "cuda":
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.model.device)
        return self._side

    def _join_side(self):
        if getattr(self, "_side_used", False):
            torch.cuda.current_stream(self.model.device).wait_stream(self._side)
            self._side_used = False

    def _loss_grad(self, buf, y, binary):
        acc = getattr(self, "_loss_acc", None)        # fit(): the epoch's running sum of per-sample losses (no per-step launches for it)
        if acc is None:
            buf["loss"].zero_()
        ops.bce_grad(buf["pred"], y, buf["dlogit"], loss_sum=buf["loss"] if acc is None else acc,
                     dlogit_sum=None if self.p_gbias is None else self.p_gbias.g, task="binary" if binary else "regression",
                     weight=getattr(self, "_sample_weight", None))

    def _deepfm_forward_backward(self, ws, buf, y, binary):
        model, sp = self.model, self.model.stage_plan
        add = []
        if sp.has_linear:
            add.append(ws["lin"])
        if sp.fm_group_names:
            add.append(ws["fm"])
            add.extend(ws["fm_extra"])      # DeepFM(fm_group=(...)): FM over further embedding groups (models/deepfm.py:53-54)
        cin = model.cin if self.p_cin_f else None
        if cin is not None:             # xDeepFM (models/xdeepfm.py:52-66): CIN over the embeddings -> Dense(1) -> extra logit
            nf, dim = len(sp.fields), model.cin_dim
            filt = [f.reshape(-1, f.shape[-1]) for f in cin.filters]
            ops.cin(ws["dnn_in"], filt, cin.biases, list(cin.layer_size), cin.split_half, cin.activation, fields=nf, dim=dim,
                    out=buf["maps"], save_y=buf["cin_y"])
            ops.mlp(buf["maps"], [], [], "linear", head_w=self.p_head1.w, in_dim=model.cin_out_dim, out=buf["cin_logit"])
            add.append(buf["cin_logit"])
        self._dnn_forward(ws["dnn_in"], sp.in_dim, buf, buf["pred"], head=True, add=add, binary=binary)
        self._loss_grad(buf, y, binary)
        self._dnn_backward(ws["dnn_in"], sp.in_dim, buf, buf["dx"], dlogit=buf["dlogit"])
        if cin is not None:
            ops.dense1_bwd(buf["maps"], model.cin_out_dim, self.p_head1.w, buf["dlogit"], buf["dmaps"], self.p_head1.g)
            ops.cin_bwd(ws["dnn_in"], filt, cin.biases, list(cin.layer_size), cin.split_half, cin.activation, buf["dmaps"],
                        [p.g.reshape(-1, p.g.shape[-1]) for p in self.p_cin_f], [p.g for p in self.p_cin_b], dx=buf["dx"],
                        accumulate=True, fields=nf, dim=dim, saved_y=buf["cin_y"])

    def _afm_forward_backward(self, ws, buf, y, B, binary):
        """AFM (models/afm.py:45-58): linear logit + one AFMLayer per group (with attention), or the gather's FM term."""
        model, sp = self.model, self.model.stage_plan
        add = [ws["lin"]] if sp.has_linear else []
        if sp.fm_group_names:
            add.append(ws["fm"])
            add.extend(ws["fm_extra"])
        if "afm_out" not in buf:
            buf["afm_out"] = [torch.zeros(B, 1, dtype=torch.float32, device=model.device) for _ in model.afm_layers]
        outs = buf["afm_out"]
        for g, layer, o in zip(model.groups, model.afm_layers, outs):
            first, n, dim = sp.group_slices[g]
            ops.afm(ws["dnn_in"][:, first:], layer.w("attention_W"), layer.w("attention_b"), layer.w("projection_h"),
                    layer.w("projection_p"), fields=n, dim=dim, out=o)
        head_in = outs[0] if outs else add[0].reshape(-1, 1)
        rest = [o.reshape(-1) for o in outs[1:]] if outs else add[1:]
        ops.mlp(head_in, [], [], "linear", head_w=model._one(), add=(add + rest) if outs else rest,
                global_bias=None if self.p_gbias is None else self.p_gbias.w, sigmoid_out=binary, in_dim=1, out=buf["pred"])
        self._loss_grad(buf, y, binary)
        dx = buf["dx"]
        dx.zero_()                                   # groups outside fm_group contribute nothing to the logit
        for g, layer, ps in zip(model.groups, model.afm_layers, self.p_afm):
            first, n, dim = sp.group_slices[g]
            ops.afm_bwd(ws["dnn_in"][:, first:], n, dim, layer.w("attention_W"), layer.w("attention_b"), layer.w("projection_h"),
                        layer.w("projection_p"), buf["dlogit"], dx[:, first:], ps[0].g, ps[1].g, ps[2].g, ps[3].g)

    def _din_forward_backward(self, staged, lo, hi, ws, buf, y, B, binary):
        """DIN (models/sequence/din.py:62-96): LocalActivationUnit over [q, k, q-k, q*k] per history position -> masked weighted
        sum of the keys -> DNN over [embeddings | attention output | dense] -> Dense(1).  The unit's MLP runs on the
        materialised [B*T, 4E'] input through dctr_mlp_fwd / dctr_mlp_bwd (activations saved)."""
        model, sp = self.model, self.model.stage_plan
        la, pa = model.attention.local_att, self.p_att
        T, E = model.T, model.key_dim
        bufs = model._attention_inputs(staged, lo, hi, ws)
        q, k, m = bufs["q"], bufs["k"], bufs["m"]
        if "att_in" not in buf:
            dev = model.device
            units = [kk.shape[1] for kk in la.dnn.kernels]
            buf.update(att_in=torch.empty(B * T, 4 * E, dtype=torch.float32, device=dev),
                       d_att_in=torch.empty(B * T, 4 * E, dtype=torch.float32, device=dev),
                       att_acts=[torch.empty(B * T, n, dtype=torch.float32, device=dev) for n in units],
                       score=torch.empty(B * T, dtype=torch.float32, device=dev),
                       d_score=torch.empty(B * T, dtype=torch.float32, device=dev),
                       dk=torch.empty(B, T, E, dtype=torch.float32, device=dev))
        act = la.dnn.activation
        dice = la.dnn.dice_params()
        ops.din_att_in(q, k, buf["att_in"])
        dice_batch = None
        if act in ("dice", "Dice") and not getattr(model, "hip_dice_stored_statistics", False):
            # training-mode Dice needs the statistics of ALL B*T rows of a layer before its activation: layer by layer
            # (pre-activations by the MLP kernel as a one-layer linear net, then dctr_dice_train_fwd), head last
            dice_batch, xin, kin = [], buf["att_in"], 4 * E
            if "att_z" not in buf:
                buf["att_z"] = [torch.empty_like(t) for t in buf["att_acts"]]
            for l, (kern, bias) in enumerate(zip(la.dnn.kernels, la.dnn.biases)):
                ops.mlp(xin, [kern], [bias], "linear", in_dim=kin, out=buf["att_z"][l])
                alpha, mmean, mvar = dice[l]
                dice_batch.append(ops.dice_train_fwd(buf["att_z"][l], alpha, mmean, mvar, buf["att_acts"][l], eps=1e-9, momentum=BN_MOMENTUM))
                xin, kin = buf["att_acts"][l], kern.shape[1]
            ops.mlp(xin, [], [], "linear", head_w=pa["out_w"].w, global_bias=pa["out_b"].w, in_dim=kin, out=buf["score"])
        else:
            ops.mlp(buf["att_in"], la.dnn.kernels, la.dnn.biases, act, dice=dice, head_w=pa["out_w"].w, global_bias=pa["out_b"].w,
                    in_dim=4 * E, out=buf["score"], save_acts=buf["att_acts"])
        hist_off = sp.extra_offsets["hist"]
        softmax = bool(model.attention.weight_normalization)
        if softmax:                     # att_weight_normalization=True: masked softmax over the positions, then the sum over ALL of them
            if "att_p" not in buf:
                buf.update(att_p=torch.empty(B * T, dtype=torch.float32, device=model.device),
                           ones=torch.ones(B, T, dtype=torch.uint8, device=model.device))
            ops.din_softmax(buf["score"], m, buf["att_p"])
            ops.din_wsum(buf["att_p"], buf["ones"], k, ws["dnn_in"][:, hist_off:])
        else:
            ops.din_wsum(buf["score"], m, k, ws["dnn_in"][:, hist_off:])
        self._dnn_forward(ws["dnn_in"], sp.in_dim, buf, buf["pred"], head=True, binary=binary)
        self._loss_grad(buf, y, binary)
        dx = buf["dx"]
        self._dnn_backward(ws["dnn_in"], sp.in_dim, buf, dx, dlogit=buf["dlogit"])
        if softmax:
            ops.din_wsum_bwd(dx[:, hist_off:], buf["att_p"], buf["ones"], k, buf["d_score"], buf["dk"])
            ops.din_softmax_bwd(buf["att_p"], m, buf["d_score"], buf["d_score"], d_bias=pa["out_b"].g)
        else:
            ops.din_wsum_bwd(dx[:, hist_off:], buf["score"], m, k, buf["d_score"], buf["dk"], d_bias=pa["out_b"].g)
        ops.mlp_bwd(buf["att_in"], 4 * E, la.dnn.kernels, buf["att_acts"], act, pa["out_w"].w, buf["d_score"],
                    [p.g for p in pa["kernels"]], [p.g for p in pa["biases"]], pa["out_w"].g, dx=buf["d_att_in"],
                    biases=la.dnn.biases, dice=dice, d_dice_alpha=[p.g for p in pa["alphas"]] if pa["alphas"] else None,
                    dice_batch=dice_batch, saved_z=buf["att_z"] if dice_batch is not None else None,
                    workspace=buf.setdefault("mlp_bwd_ws_att", {}))
        ops.din_att_in_bwd(buf["d_att_in"], q, k, buf["dk"], dx, self.qcol)
        for (fc, idx, hm, col), pt in zip(bufs["key_lookups"], self.p_hist):
            if pt.g is not None:                                   # frozen history table: no scatter
                ops.embed_lookup_bwd(idx, tuple(pt.w.shape), hm, buf["dk"][:, :, col:], pt.g, touched=pt.touched)

    def _nfm_forward_backward(self, ws, buf, y, binary):
        """NFM (models/nfm.py:49-58): DNN over [BiInteractionPooling(embeddings) (+ Dropout(bi_dropout)) | dense] -> Dense(1) + linear logit."""
        model, sp = self.model, self.model.stage_plan
        off = sp.extra_offsets["bi_interaction"]
        x = ws["dnn_in"][:, off:]
        E = model.emb_dim
        bi_rate = float(getattr(model, "bi_dropout", 0) or 0)
        ops.bi_interaction(ws["dnn_in"], fields=model.n_emb, dim=E, out=x)
        if bi_rate > 0:                 # nfm.py:52-53: Dropout on the pooled vector, in place (layer slot 100 of the step's mask seeds)
            xe = x[:, :E]
            ops.dnn_train_layer(xe, "linear", h=xe, dropout_rate=bi_rate, dropout_seed=self.dropout_seed(100))
        self._dnn_forward(x, model.dnn_in_dim, buf, buf["pred"], head=True, add=[ws["lin"]] if sp.has_linear else [], binary=binary)
        self._loss_grad(buf, y, binary)
        dx = buf["dx"]
        self._dnn_backward(x, model.dnn_in_dim, buf, dx[:, off:], dlogit=buf["dlogit"])
        dy = dx[:, off:]
        if bi_rate > 0:
            if "bi_dz" not in buf:
                buf["bi_dz"] = torch.empty(x.shape[0], E, dtype=torch.float32, device=model.device)
            dy = ops.dnn_train_layer(x[:, :E], "linear", dropout_rate=bi_rate, dropout_seed=self.dropout_seed(100), dh=dx[:, off:off + E],
                                     dz=buf["bi_dz"])
        ops.bi_interaction_bwd(ws["dnn_in"], model.n_emb, E, dy, dx)

    def _pnn_forward_backward(self, ws, buf, y, binary):
        """PNN, inner-product form (models/pnn.py:52-72): DNN over [embeddings | pair inner products | dense] -> Dense(1)."""
        model, sp = self.model, self.model.stage_plan
        if model.use_inner:
            off = sp.extra_offsets["inner_product"]
            ops.inner_product(ws["dnn_in"], True, fields=model.n_emb, dim=model.emb_dim, out=ws["dnn_in"][:, off:])
        self._dnn_forward(ws["dnn_in"], sp.in_dim, buf, buf["pred"], head=True, binary=binary)
        self._loss_grad(buf, y, binary)
        dx = buf["dx"]
        self._dnn_backward(ws["dnn_in"], sp.in_dim, buf, dx, dlogit=buf["dlogit"])
        if model.use_inner:
            ops.inner_product_bwd(ws["dnn_in"], model.n_emb, model.emb_dim, dx[:, off:], dx, accumulate=True)

    def _dcn_forward_backward(self, ws, buf, y, B, binary):
        """DCN (models/dcn.py:45-78): [CrossNet(dnn_in), DNN(dnn_in)] -> Dense(1) + linear logit -> PredictionLayer."""
        model, sp = self.model, self.model.stage_plan
        d = sp.in_dim
        stack = model._stack.get(B)
        if stack is None:
            stack = model._stack[B] = torch.zeros(B, (model.width + 3) // 4 * 4, dtype=torch.float32, device=model.device)
        col = 0
        par = getattr(model.cross, "parameterization", "vector") if model.cross is not None else "vector"
        if model.cross is not None:
            if self.p_mix is not None:
                ops.crossnet_mix(ws["dnn_in"], *[p.w for p in self.p_mix], dim=d, out=stack)
            else:
                self._cross_fwd(ws["dnn_in"], d, par, stack, buf)
            col = d
        if model.dnn is not None:
            self._dnn_forward(ws["dnn_in"], d, buf, stack[:, col:], head=False)
        add = [ws["lin"]] if sp.has_linear else []
        ops.mlp(stack, [], [], "linear", head_w=self.p_head.w, add=add,
                global_bias=None if self.p_gbias is None else self.p_gbias.w, sigmoid_out=binary, in_dim=model.width,
                out=buf["pred"])
        self._loss_grad(buf, y, binary)
        dstack = buf["dstack"]
        ops.dense1_bwd(stack, model.width, self.p_head.w, buf["dlogit"], dstack, self.p_head.g)
        have_dx = False
        if model.dnn is not None:
            self._dnn_backward(ws["dnn_in"], d, buf, buf["dx"], d_out=dstack[:, col:])
            have_dx = True
        if model.cross is not None and self.p_mix is not None:
            ops.crossnet_mix_bwd(ws["dnn_in"], d, [p.w for p in self.p_mix], dstack, [p.g for p in self.p_mix], buf["dx"],
                                 accumulate=have_dx)
        elif model.cross is not None:
            ops.crossnet_bwd(ws["dnn_in"], d, self.p_cross_k.w, self.p_cross_b.w, par, dstack, self.p_cross_k.g, self.p_cross_b.g,
                             buf["dx"], accumulate=have_dx, saved_u=buf.get("cross_u") if par == "matrix" else None,
                             saved_x=buf.get("cross_x") if par == "matrix" else None)

    def _cross_fwd(self, dnn_in, d, par, stack, buf=None):
        import ctypes
        from . import _C
        mode = _C.CROSS_VECTOR if par == "vector" else _C.CROSS_MATRIX
        ks, bs = self.p_cross_k.w, self.p_cross_b.w
        need = int(_C.lib().dctr_crossnet_workspace_bytes(d, ks.shape[0], mode, ctypes.c_void_p(ks.data_ptr())))
        if need and (getattr(self, "_cross_ws", None) is None or self._cross_ws.numel() * 4 < need):
            self._cross_ws = torch.empty(need // 4, dtype=torch.float32, device=ks.device)
        su = sx = None
        if par == "matrix" and buf is not None and ks.shape[0] >= 1:
            # the forward kernel writes u_l = W_l x_l and x_1 .. x_{L-1} for the backward (dctr_crossnet_bwd_args_t.saved_u / saved_x:
            # no recompute GEMM + elementwise launch per layer there)
            L, B = ks.shape[0], dnn_in.shape[0]
            if "cross_u" not in buf:
                buf["cross_u"] = torch.empty(L, B, d, dtype=torch.float32, device=ks.device)
                buf["cross_x"] = torch.empty(max(L - 1, 1), B, d, dtype=torch.float32, device=ks.device)
            su, sx = buf["cross_u"], buf["cross_x"]
        a = _C.CrossnetArgs(x=dnn_in.data_ptr(), batch=dnn_in.shape[0], x_stride=dnn_in.stride(0), dim=d, layers=ks.shape[0], mode=mode,
                            workspace_ready=0, kernels=ks.data_ptr(), bias=bs.data_ptr(), y=stack.data_ptr(), y_stride=stack.stride(0),
                            workspace=self._cross_ws.data_ptr() if need else None, workspace_bytes=need,
                            save_u=None if su is None else su.data_ptr(), save_x=None if sx is None else sx.data_ptr())
        if su is not None:
            # the one-kernel form keeps [16, dim] tiles of x_0 / x_l / x_{l+1} in LDS: whether it takes this width is the library's answer
            key = (dnn_in.shape[0], d, ks.shape[0])
            ok = self._cross_one_kernel.get(key)
            if ok is None:
                if len(self._cross_one_kernel) > 64:            # (ragged batch sizes must not pile up)
                    self._cross_one_kernel.clear()
                ok = self._cross_one_kernel[key] = bool(_C.lib().dctr_crossnet_fwd_supported(ctypes.byref(a), None))
            if not ok:
                return self._cross_fwd_layered(dnn_in, d, stack, su, sx)
        _C.check(_C.lib().dctr_crossnet_head_fwd(ctypes.byref(a), _C.stream_ptr()), "dctr_crossnet_head_fwd")

    def _cross_fwd_layered(self, dnn_in, d, stack, su, sx):
        """Matrix CrossNet (interaction.py:416-420) of any width, layer by layer, writing what dctr_crossnet_bwd reads: u_l = x_l W_l^T
        (the library's f32-MFMA GEMM, dctr_sgemm) -> saved_u[l]; x_{l+1} = x_0 * (u_l + b_l) + x_l (dctr_crossnet_matrix_step) ->
        saved_x[l] (x_1 .. x_{L-1}) and, for the last layer, the cross half of the stack."""
        import ctypes  # noqa: F401
        from . import _C
        ks, bs = self.p_cross_k.w, self.p_cross_b.w       # [L, d, d] (W_l: [out n, in k]), [L, d]
        L, B, st = ks.shape[0], dnn_in.shape[0], _C.stream_ptr()
        xl, xl_stride = dnn_in, ops.row_stride(dnn_in)
        for l in range(L):
            u = su[l]
            # column-major BLAS view: u^T (d x B) = W^T-view (k x n)^T . x_l^T (k x B)
            _C.check(_C.lib().dctr_sgemm(1, 0, d, B, d, ks[l].data_ptr(), d, 0, xl.data_ptr(), int(xl_stride), 0, 0.0, u.data_ptr(), d, 0, 1, st),
                     "dctr_sgemm")
            nxt, nxt_stride = (stack, ops.row_stride(stack)) if l == L - 1 else (sx[l], d)
            _C.check(_C.lib().dctr_crossnet_matrix_step(dnn_in.data_ptr(), ops.row_stride(dnn_in), xl.data_ptr(), int(xl_stride), u.data_ptr(),
                                                        bs[l].data_ptr(), B, d, nxt.data_ptr(), int(nxt_stride), st), "dctr_crossnet_matrix_step")
            xl, xl_stride = nxt, nxt_stride

    def step(self, staged, lo, hi, y, apply=True, loss_acc=None, weight=None):
        """One optimizer step on rows [lo, hi) of the staged inputs; y: device float tensor [hi-lo].  Returns the mean
        loss of the batch BEFORE the update (a device tensor; no host synchronisation here).  ``apply=False`` stops
        after the backward pass and leaves the gradients in the ``g`` buffers (tests).  ``loss_acc`` (a device float32 tensor of
        one element): the batch's SUMMED loss is added to it instead — no per-step zero / divide launches, nothing returned
        (fit() keeps one accumulator per epoch).  ``weight``: device float32 [hi-lo], tf.keras' per-sample weights of the batch
        (loss = sum w_b l_b / B: dctr_bce_grad_w)."""
        self._loss_acc = loss_acc
        self._sample_weight = weight
        model, sp = self.model, self.model.stage_plan
        model._trainer_owns_cross = self.is_dcn            # (_DCN._begin: no re-packing of the cross weights for this call)
        model._trainer_step = True          # (_begin: only what the step reads — no inference-form BatchNormalization scale / shift, no
        try:                                #  zero-padded DNN copies, no packed cross operands; predict() refreshes them itself)
            model._begin()                  # weight-derived forward buffers follow the last update
        finally:
            model._trainer_owns_cross = False
            model._trainer_step = False
        self.n_steps += 1
        B = hi - lo
        buf = self._buffers(B)
        binary = model.task == "binary"
        # forward (pool launches + two launches, activations saved)
        sp.pool_trace = []
        try:
            ws = sp.run(staged, lo, hi)
            pool_calls = sp.pool_trace
        finally:
            sp.pool_trace = None
        if self.lin_only_params:            # the linear-only features' logit joins the linear logit every model adds to its head
            ws["lin"].add_(ws["lin2"])
        if self.is_dcn:
            self._dcn_forward_backward(ws, buf, y, B, binary)
        elif self.is_din:
            self._din_forward_backward(staged, lo, hi, ws, buf, y, B, binary)
        elif self.is_afm:
            self._afm_forward_backward(ws, buf, y, B, binary)
        elif self.is_nfm:
            self._nfm_forward_backward(ws, buf, y, binary)
        elif self.is_pnn:
            self._pnn_forward_backward(ws, buf, y, binary)
        else:
            self._deepfm_forward_backward(ws, buf, y, binary)
        # embedding / linear / FM backward
        for t in list(buf["pooled_g"].values()) + list(buf["pooled_lin_g"].values()):
            t.zero_()
        for gname in sp.fm_group_names[1:]:      # FM groups beyond the gather's own: d e_f += dlogit (S_g - e_f) on their slice of dnn_in
            first, nf_g, dim_g = sp.group_slices[gname]
            ops.fm_bwd(ws["dnn_in"][:, first:], nf_g, dim_g, buf["dlogit"], buf["dx"][:, first:], accumulate=True)
        g = sp.gather_args(staged, lo, hi, ws)
        ops.embed_gather_fm_bwd(g, buf["field_grads"], d_dnn_in=buf["dx"], d_fm=buf["dlogit"] if sp.fm_group_names else None,
                                d_lin=buf["dlogit"] if sp.has_linear else None,
                                g_dense_lin_w=None if self.p_dense_lin is None else self.p_dense_lin.g,
                                dense_lin_rows=self.dense_rows)
        # sequence features: pooled-vector gradients -> rows of their tables
        pooled = [(f, pt, pl) for f, pt, pl in self.field_params if f.kind == "pooled"]
        assert len(pool_calls) == len(pooled) + len(self.lin_only_varlen)
        for (args, _keep), (f, pt, pl) in zip(pool_calls, pooled):
            ops.embed_pool_bwd(args, d_out=buf["pooled_g"][f.fc.name], d_lin_out=buf["pooled_lin_g"].get(f.fc.name),
                               g_table=pt.g, g_lin_table=None if pl is None else pl.g, touched=pt.touched)
        if self.lin_only_params:
            # features of the linear part alone: d logit -> rows of their linear tables (the backward of run_lin_only's gather; pooled
            # sequences through their per-batch vectors)
            for t in buf["lin2_pool_g"].values():
                t.zero_()
            nf = len(sp.fields)
            g2 = ops.make_gather_args(ws["desc2"], ws["n_fields2"], staged.ids[nf:, lo:hi], staged.ids.stride(0), 1, B, 1, False,
                                      ws["any_hash2"], lin_logit=ws["lin2"], status=ws["status"])
            ops.embed_gather_fm_bwd(g2, buf["field_grads2"], d_lin=buf["dlogit"])
            for (args, _keep), (fc, p) in zip(pool_calls[len(pooled):], self.lin_only_varlen):
                if p.g is not None:
                    ops.embed_pool_bwd(args, d_out=buf["lin2_pool_g"][fc.name], g_table=p.g)
        self._join_side()
        if apply:
            self.apply_update()
        return None if loss_acc is not None else buf["loss"] / B

    def apply_update(self):
        """The optimizer step over every parameter in one launch (the gradients are zeroed behind it).  Separate from ``step`` so
        that a data-parallel fit can exchange the gradients of a step between its backward and its update (training._DataParallel)."""
        self.t += 1
        lr = self.lr
        if self.kind == "adam":
            lr = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        # fit(): the l2 penalties of THIS step's weights, weighted by the batch's rows, ride the launch (tf.keras' reported loss is the
        # batch-size-weighted mean of data loss + regularisation losses: training._fit_hip sets penalty_acc / penalty_rows)
        pen = self.penalty_acc if self.penalty_rows else None
        ops.opt_multi(self.kind, self.segs, self.n_segs, self.max_n, lr, self.b1, self.b2, self.eps, penalty=pen,
                      penalty_scale=float(self.penalty_rows) if pen is not None else 0.0)
        # the weights moved through raw pointers (torch's version counters did not): derived inference copies are stale
        self.model._raw_weight_writes = getattr(self.model, "_raw_weight_writes", 0) + 1

    def batch_statistics(self):
        """True when the step takes statistics over the BATCH (training-mode BatchNormalization / Dice): under data-parallel fit a rank
        normalises with its own sub-batch's statistics (training._DataParallel: per-replica statistics, stored ones averaged)."""
        la = self.model.attention.local_att if self.is_din else None
        din_dice = la is not None and la.dnn.activation in ("dice", "Dice") and not getattr(self.model, "hip_dice_stored_statistics", False)
        return bool(self.bn_layers or self.dice_dnn or din_dice)
