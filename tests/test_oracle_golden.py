"""CPU: the stand-alone NumPy oracle (oracle/ref_numpy.py, oracle/ref_models.py) against the golden
fixtures produced by executing the reference's own Python (oracle/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import farmhash as fh
from oracle import ref_models as RM
from oracle import ref_numpy as R
from tests.util import assert_close, golden_meta, load_golden, sigmoid_inv
from tests.spec import columns_from_spec


# ---------------------------------------------------------------------------------------------
# hash
# ---------------------------------------------------------------------------------------------
KATS = [  # TensorFlow frozen vectors / pyfarmhash README — see oracle/farmhash64.c header
    (b"Hello", 15404698994557526151), (b"World", 18308117990299812472),
    (b"a", 12917804110809363939), (b"b", 11795596070477164822), (b"c", 11430444447143000872),
    (b"d", 4470636696479570465), (b"abc", 2640714258260161385),
]


@pytest.mark.parametrize("s,expect", KATS)
def test_fingerprint64_known_answers(s, expect):
    assert fh.fingerprint64(s) == expect
    assert fh.fingerprint64_py(s) == expect


def test_to_hash_bucket_fast_doc_examples():
    assert [fh.fingerprint64(s) % 3 for s in (b"Hello", b"TensorFlow", b"2.x")] == [0, 2, 2]   # TF API doc
    assert [fh.fingerprint64(s) % 10 for s in (b"a", b"b", b"c", b"d")] == [9, 2, 2, 5]        # TF op test
    assert [fh.fingerprint64(s) % 3 for s in (b"A", b"B", b"C", b"D", b"E")] == [1, 0, 1, 1, 2]  # keras Hashing doc


def test_c_and_python_restatements_agree_on_every_length_branch():
    rng = np.random.RandomState(0)
    for n in list(range(0, 140)) + [255, 256, 257, 1000]:
        s = bytes(rng.randint(0, 256, n).astype(np.uint8))
        assert fh.fingerprint64(s) == fh.fingerprint64_py(s), n


def _abseil_cityhash64():
    """abseil's CityHash64 as compiled into pyarrow's libarrow (a local symbol: called through its address).  An
    implementation neither written nor built by this repository."""
    import ctypes
    import glob
    import shutil
    import subprocess
    try:
        import pyarrow
    except ImportError:
        return None
    nm = shutil.which("nm")
    libs = sorted(glob.glob(os.path.join(os.path.dirname(pyarrow.__file__), "libarrow.so*")))
    if nm is None or not libs:
        return None
    out = subprocess.run([nm, libs[0]], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True).stdout
    addr = [int(line.split()[0], 16) for line in out.splitlines() if line.endswith("hash_internal10CityHash64EPKcm")]
    if not addr:
        return None
    lib = ctypes.CDLL(libs[0])
    base = min(int(line.split("-")[0], 16) for line in open("/proc/self/maps") if libs[0] in line)
    fn = ctypes.CFUNCTYPE(ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t)(base + addr[0])
    fn._keep = lib
    return fn


def test_fingerprint64_against_an_independent_cityhash64_up_to_32_bytes():
    """External pin of the <= 16 B and the 17-32 B branches of Fingerprint64 (the latter is what int64 ids >= 10^16 take as
    decimal strings, reference layers/utils.py:103-107): farmhashna::Hash64 is CityHash64 v1.1 for inputs of up to 32 bytes
    (farmhash.cc keeps HashLen0to16 / HashLen17to32 unchanged; from 33 bytes on the two differ), and abseil carries CityHash64
    v1.1.  Both restatements (C and Python) and the hash-bucket assignment built on them are checked against it."""
    city = _abseil_cityhash64()
    if city is None:
        pytest.skip("no abseil CityHash64 found in this image (pyarrow's libarrow + nm)")
    rng = np.random.RandomState(5)
    cases = [bytes(rng.randint(0, 256, n).astype(np.uint8)) for n in range(0, 33) for _ in range(40)]
    ids = [10 ** 16, 10 ** 16 + 1, 2 ** 63 - 1, -(2 ** 63), -10 ** 17, 123456789012345678, 99999999999999999, 10 ** 18 + 7]
    ids += [int(x) for x in rng.randint(10 ** 16, 2 ** 62, 200, dtype=np.int64)] + [-int(x) for x in rng.randint(10 ** 15, 2 ** 62, 100, dtype=np.int64)]
    cases += [str(i).encode() for i in ids]
    assert all(17 <= len(str(i)) <= 20 for i in ids)
    for s in cases:
        want = city(s, len(s))
        assert fh.fingerprint64_py(s) == want, (s, "python restatement")
        assert fh.fingerprint64(s) == want, (s, "C restatement")
    # ... and the bucket assignment of such ids, as Hash.call computes it: Fingerprint64(decimal string) mod num_buckets
    for nb, mz in ((1000003, False), (97, True)):
        got = fh.hash_bucket_int(np.asarray(ids, dtype=np.int64), nb, mask_zero=mz)
        want = [city(str(i).encode(), len(str(i))) % (nb - 1 if mz else nb) + (1 if mz else 0) for i in ids]
        assert [int(g) for g in got] == want
    # sanity of the comparison itself: from 33 bytes on CityHash64 and Fingerprint64 are different functions
    s = bytes(range(40))
    assert city(s, 40) != fh.fingerprint64(s)


def test_reference_vocabulary_vector():
    """The reference's only known-answer assertion on this path: tests/layers/utils_test.py:15-33."""
    g = load_golden("hash")
    path = os.path.join(os.path.dirname(__file__), "golden", "_vocabulary_example.csv")
    with open(path, "wb") as f:
        f.write(bytes(g["vocab_csv"]))
    keys = np.array([k.decode() for k in g["vocab_keys"]], dtype=object).reshape(-1, 1)
    got = fh.hash_layer(keys, 4, False, vocabulary_path=path)
    assert got.reshape(-1).tolist() == [1, 3, 0] == g["vocab_expected"].tolist()
    os.remove(path)


def test_hash_layer_matches_reference_code():
    g = load_golden("hash")
    strs = np.array([s.decode() for s in g["strs"]], dtype=object)
    for nb in (4, 1000, 100000, 2 ** 31 - 1):
        for mz in (0, 1):
            assert (fh.hash_layer(g["ints32"], nb, bool(mz)) == g["i32_nb%d_mz%d" % (nb, mz)]).all()
            assert (fh.hash_layer(g["ints64"], nb, bool(mz)) == g["i64_nb%d_mz%d" % (nb, mz)]).all()
            assert (fh.hash_layer(strs, nb, bool(mz)) == g["str_nb%d_mz%d" % (nb, mz)]).all()
    t = load_golden("criteo_tokens")
    toks = np.array([s.decode() for s in t["tokens"]], dtype=object)
    assert (fh.hash_layer(toks, 1000, False) == t["hash_nb1000"]).all()


# ---------------------------------------------------------------------------------------------
# interaction layers
# ---------------------------------------------------------------------------------------------
def test_fm():
    g = load_golden("interaction")
    for tag in ("t", "c2", "one"):
        assert_close(R.fm(g["fm_%s_x" % tag]), g["fm_%s_y" % tag], what="fm " + tag)
    # analytic identity: FM == sum_{i<j} <e_i, e_j>
    x = g["fm_c2_x"].astype(np.float64)
    brute = sum((x[:, i] * x[:, j]).sum(-1) for i in range(x.shape[1]) for j in range(i + 1, x.shape[1]))
    assert_close(R.fm(x)[:, 0], brute, rtol=1e-10, atol=1e-12, what="fm identity")


def test_cin():
    g = load_golden("interaction")
    meta = golden_meta(g)
    for tag in "abcde":
        m = meta["cin_" + tag]
        n = len(m["layer_size"])
        y = R.cin(g["cin_%s_x" % tag], [g["cin_%s_filter%d" % (tag, k)] for k in range(n)],
                  [g["cin_%s_bias%d" % (tag, k)] for k in range(n)], m["split_half"], m["activation"])
        assert_close(y, g["cin_%s_y" % tag], what="cin " + tag)


def test_cin_bruteforce_triple_loop():
    rng = np.random.RandomState(0)
    B, F0, D, H = 2, 3, 2, 4
    x = rng.standard_normal((B, F0, D))
    w = rng.standard_normal((1, F0 * F0, H))
    b = rng.standard_normal(H)
    got = R.cin(x, [w], [b], split_half=False, activation="linear")
    want = np.zeros((B, H))
    for bb in range(B):
        for h in range(H):
            for d in range(D):
                acc = b[h]
                for i in range(F0):
                    for j in range(F0):
                        acc += x[bb, i, d] * x[bb, j, d] * w[0, i * F0 + j, h]
                want[bb, h] += acc
    assert_close(got, want, rtol=1e-10, atol=1e-12, what="cin brute force")


def test_crossnet():
    g = load_golden("interaction")
    meta = golden_meta(g)
    for tag in ("v0", "v1", "v3", "m1", "m2", "v2w", "m2w"):
        m = meta["cross_" + tag]
        n = m["layer_num"]
        y = R.crossnet(g["cross_%s_x" % tag], [g["cross_%s_kernel%d" % (tag, k)] for k in range(n)],
                       [g["cross_%s_bias%d" % (tag, k)] for k in range(n)], m["parameterization"])
        assert_close(y, g["cross_%s_y" % tag], what="crossnet " + tag)


def mix_weights(g, tag):
    """(U_list, V_list, C_list, gating kernels, biases) of a CrossNetMix fixture, in layer / expert order."""
    meta = golden_meta(g)[tag]
    pre = "mix_%s_w/" % tag
    W = lambda k: [g["%scross_net_mix/%s%d" % (pre, k, i)] for i in range(meta["layer_num"])]   # noqa: E731
    gating = [g[pre + ("dense/kernel" if e == 0 else "dense_%d/kernel" % e)] for e in range(meta["num_experts"])]
    return W("U_list"), W("V_list"), W("C_list"), gating, W("bias")


def test_crossnet_mix():
    g = load_golden("crossnet_mix")
    for tag in ("a", "b", "c"):
        U, V, C, gating, bias = mix_weights(g, tag)
        y = R.crossnet_mix(g["mix_%s_x" % tag], U, V, C, gating, bias)
        assert_close(y, g["mix_%s_y" % tag], what="crossnet_mix " + tag)
        y64 = R.crossnet_mix(g["mix_%s_x" % tag].astype(np.float64), *[[w.astype(np.float64) for w in ws]
                                                                      for ws in (U, V, C, gating, bias)])
        assert_close(y64, g["mix_%s_y" % tag], what="crossnet_mix f64 " + tag)


def test_afm_inner_product():
    g = load_golden("interaction")
    for tag in ("t", "w", "two"):
        x = g["afm_%s_x" % tag]
        embeds = [x[:, i:i + 1, :] for i in range(x.shape[1])]
        y = R.afm(embeds, g["afm_%s_attention_W" % tag], g["afm_%s_attention_b" % tag], g["afm_%s_projection_h" % tag],
                  g["afm_%s_projection_p" % tag])
        assert_close(y, g["afm_%s_y" % tag], what="afm " + tag)
        assert_close(R.inner_product(embeds, True), g["ip_%s_sum" % tag], what="ip sum " + tag)
        assert_close(R.inner_product(embeds, False), g["ip_%s_full" % tag], what="ip full " + tag)
    # identity: FM == sum of InnerProduct(reduce_sum)
    x = g["afm_w_x"].astype(np.float64)
    embeds = [x[:, i:i + 1, :] for i in range(x.shape[1])]
    assert_close(R.inner_product(embeds, True).sum(axis=(1, 2)), R.fm(x)[:, 0], rtol=1e-10, atol=1e-12)


# ---------------------------------------------------------------------------------------------
# sequence layers
# ---------------------------------------------------------------------------------------------
def test_sequence_pooling_and_weighting():
    g = load_golden("sequence")
    seq, lengths, mask, w = g["seq"], g["lengths"], g["mask"], g["w"]
    for mode in ("sum", "mean", "max"):
        assert_close(R.sequence_pooling(seq, mode, lengths=lengths), g["pool_len_" + mode], what="pool len " + mode)
        assert_close(R.sequence_pooling(seq, mode, mask=mask), g["pool_mask_" + mode], what="pool mask " + mode)
    for wn in (0, 1):
        assert_close(R.weighted_sequence(seq, w, lengths=lengths, weight_normalization=bool(wn)), g["wseq_len_wn%d" % wn])
        assert_close(R.weighted_sequence(seq, w, mask=mask, weight_normalization=bool(wn)), g["wseq_mask_wn%d" % wn])


def _att_weights(g, prefix, n_layers, act):
    ks = [g["%s/dnn/kernel%d" % (prefix, i)] for i in range(n_layers)]
    bs = [g["%s/dnn/bias%d" % (prefix, i)] for i in range(n_layers)]
    dice = None
    if act == "dice":
        dice = []
        for i in range(n_layers):
            sfx = "" if i == 0 else "_%d" % i
            dice.append((g["%s/dice%s/dice_alpha" % (prefix, sfx)], g["%s/batch_normalization%s/moving_mean" % (prefix, sfx)],
                         g["%s/batch_normalization%s/moving_variance" % (prefix, sfx)]))
    return ks, bs, g[prefix + "/local_activation_unit/kernel"], g[prefix + "/local_activation_unit/bias"], dice


def test_attention_sequence_pooling():
    g = load_golden("sequence")
    meta = golden_meta(g)
    seq, lengths, mask, query = g["seq"], g["lengths"], g["mask"], g["query"]
    len_mask = R.sequence_mask(lengths, seq.shape[1])
    for tag in ("sig", "sig_wn", "dice", "dice_wn", "relu"):
        m = meta["att_" + tag]
        for form, km in (("len", len_mask), ("mask", mask)):
            ks, bs, ok, ob, dice = _att_weights(g, "att_%s_%s_w" % (tag, form), len(m["hidden"]), m["activation"])
            y = R.attention_sequence_pooling(query, seq, km, ks, bs, ok, ob, m["activation"], dice,
                                             m["weight_normalization"])
            assert_close(y, g["att_%s_%s_y" % (tag, form)], what="attention %s %s" % (tag, form))


def test_core_layers():
    g = load_golden("core")
    x = g["x"]
    for tag, n, act in (("relu", 3, "relu"), ("dice", 2, "dice"), ("sig", 1, "sigmoid")):
        pre = "dnn_%s_w/dnn" % tag
        ks = [g["%s/kernel%d" % (pre, i)] for i in range(n)]
        bs = [g["%s/bias%d" % (pre, i)] for i in range(n)]
        dice = None
        if act == "dice":
            dice = []
            for i in range(n):
                sfx = "" if i == 0 else "_%d" % i
                dice.append((g["dnn_dice_w/dice%s/dice_alpha" % sfx], g["dnn_dice_w/batch_normalization%s/moving_mean" % sfx],
                             g["dnn_dice_w/batch_normalization%s/moving_variance" % sfx]))
        assert_close(R.dnn(x, ks, bs, act, dice), g["dnn_%s_y" % tag], what="dnn " + tag)
    for task in ("binary", "regression"):
        assert_close(R.prediction_layer(g["logit"], g["pred_%s_bias" % task], task), g["pred_%s_y" % task])
    assert_close(R.linear(g["lin_sparse"]), g["lin_mode0_y"])
    assert_close(R.linear(None, g["lin_dense"], g["lin_mode1_kernel"], g["lin_mode1_bias"]), g["lin_mode1_y"])
    assert_close(R.linear(g["lin_sparse"], g["lin_dense"], g["lin_mode2_kernel"]), g["lin_mode2_y"])


# ---------------------------------------------------------------------------------------------
# whole models (reference constructors executed over the shim)
# ---------------------------------------------------------------------------------------------
MODEL_FIXTURES = ["model_deepfm_mixed", "model_deepfm_hash", "model_dcn_vector", "model_dcn_matrix", "model_dcn_crossonly",
                  "model_xdeepfm", "model_xdeepfm_nosplit", "model_din_ref_dice_hash0", "model_din_ref_sigmoid_hash0",
                  "model_din_ref_dice_hash1", "model_din_ref_sigmoid_hash1", "model_din_big_wn0", "model_din_big_wn1",
                  "model_deepfm_criteo_sample", "model_wdl", "model_wdl_wide_subset", "model_fnn", "model_wdl_fixed",
                  "model_fnn_fixed", "model_afm", "model_afm_two_groups", "model_afm_noatt", "model_pnn_inner",
                  "model_pnn_plain", "model_nfm", "model_nfm_fixed", "model_dcnmix", "model_dcnmix_crossonly",
                  "model_dcnmix_fixed", "model_deepfm_bn", "model_dcn_bn", "model_xdeepfm_bn", "model_deepfm_bn_fixed",
                  "model_din_bn_dice", "model_din_bn_sigmoid", "model_xdeepfm_d12", "model_deepfm_auto", "model_wdl_e80"]


def run_oracle_model(g, dtype=np.float32, task=None, abs_weights=False):
    """task: override the fixture's task ('regression' = the logit PredictionLayer receives); abs_weights: every weight replaced by
    its magnitude — for models made of sums, products and ReLU an upper bound of the magnitude every sum is taken at
    (tests/util.assert_close_terms)."""
    meta = golden_meta(g)
    weights = {k[2:]: (np.abs(v) if abs_weights else v) for k, v in g.items() if k.startswith("w/")}
    feed = {k[5:]: v for k, v in g.items() if k.startswith("feed/")}
    dnn_cols = columns_from_spec(meta["dnn"])
    lin_cols = columns_from_spec(meta["linear"])
    kw = dict(meta["kwargs"])
    kw["dtype"] = dtype
    if task is not None:
        kw["task"] = task
    name = meta["model"]
    if name == "DeepFM":
        return RM.deepfm(lin_cols, dnn_cols, weights, feed, **kw)
    if name == "DCN":
        return RM.dcn(lin_cols, dnn_cols, weights, feed, **kw)
    if name == "xDeepFM":
        return RM.xdeepfm(lin_cols, dnn_cols, weights, feed, **kw)
    if name == "DIN":
        return RM.din(dnn_cols, meta["extra_args"][0], weights, feed, **kw)
    if name == "WDL":
        return RM.wdl(lin_cols, dnn_cols, weights, feed, **kw)
    if name == "FNN":
        return RM.fnn(lin_cols, dnn_cols, weights, feed, **kw)
    if name == "AFM":
        return RM.afm(lin_cols, dnn_cols, weights, feed, **kw)
    if name == "PNN":
        return RM.pnn(lin_cols, dnn_cols, weights, feed, **kw)
    if name == "DCNMix":
        return RM.dcnmix(lin_cols, dnn_cols, weights, feed, **kw)
    if name == "NFM":
        return RM.nfm(lin_cols, dnn_cols, weights, feed, **kw)
    raise KeyError(name)


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_model_oracle_matches_reference_code(name):
    g = load_golden(name)
    y = run_oracle_model(g)
    ref = g["y"]
    assert y.shape == ref.shape
    # compare on the probability and (where not saturated) on the logit
    assert_close(y, ref, rtol=1e-4, atol=1e-6, what=name + " prob")
    ok = (ref > 1e-6) & (ref < 1 - 1e-6)
    if ok.any():
        assert_close(sigmoid_inv(y[ok]), sigmoid_inv(ref[ok]), rtol=1e-4, atol=2e-5, what=name + " logit")


@pytest.mark.parametrize("name,kw", [("dnn_relu_sigmoid_bn", dict(activation="relu", output_activation="sigmoid", use_bn=True)),
                                     ("dnn_tanh_linear", dict(activation="tanh", output_activation="linear", use_bn=False))])
def test_dnn_layer_with_bn_and_output_activation(name, kw):
    """DNN(use_bn, output_activation) of the reference's own layer code (layers/core.py:176-184,189-208) vs the restatement."""
    g = load_golden(name)
    n = 3
    ks = [g["w/dnn/kernel%d" % i] for i in range(n)]
    bs = [g["w/dnn/bias%d" % i] for i in range(n)]
    bn = None
    if kw["use_bn"]:
        bn = [tuple(g["w/batch_normalization%s/%s" % ("" if i == 0 else "_%d" % i, w)] for w in ("gamma", "beta", "moving_mean", "moving_variance"))
              for i in range(n)]
    y = R.dnn(g["x"], ks, bs, kw["activation"], output_activation=kw["output_activation"], bn_params=bn)
    assert_close(y, g["y"], rtol=1e-5, atol=1e-6, what=name)


# ---------------------------------------------------------------------------------------------
# the recipe itself
# ---------------------------------------------------------------------------------------------
@pytest.mark.skipif(not os.path.isdir("/root/reference/deepctr"), reason="needs /root/reference (build container only)")
def test_recipe_regenerates_every_fixture(tmp_path):
    """`python -m oracle.make_golden` (every generator, through main()) into a scratch directory: every committed fixture comes
    out again byte for byte, and nothing is committed that the recipe does not produce (main() had rotted)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "golden")
    r = subprocess.run([sys.executable, "-m", "oracle.make_golden", "--out", out], cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    committed = sorted(os.listdir(os.path.join(root, "tests", "golden")))
    assert sorted(os.listdir(out)) == committed
    for name in committed:
        a = open(os.path.join(root, "tests", "golden", name), "rb").read()
        b = open(os.path.join(out, name), "rb").read()
        if a == b:
            continue
        assert name.endswith(".npz"), name + " differs"
        za, zb = np.load(os.path.join(root, "tests", "golden", name)), np.load(os.path.join(out, name))   # (container bytes may differ with zlib)
        assert sorted(za.files) == sorted(zb.files), name
        for k in za.files:
            assert za[k].dtype == zb[k].dtype and za[k].shape == zb[k].shape and za[k].tobytes() == zb[k].tobytes(), "%s[%s]" % (name, k)
