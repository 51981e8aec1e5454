"""CPU: the ``deepctr`` facade (reference import name over this build) — module paths of SURVEY.md §8(b), object identity with
``deepctr_amd``, no import-time network/thread side effects (reference deepctr/__init__.py:1-4 starts a version-check thread)."""
import importlib
import sys
import threading


def test_facade_module_paths_and_identity():
    before = threading.active_count()
    import deepctr
    import deepctr_amd.feature_column
    import deepctr_amd.layers
    import deepctr_amd.models
    assert threading.active_count() == before, "importing deepctr must not start a version-check thread"
    assert "requests" not in [m for m in sys.modules if m == "requests" and getattr(sys.modules[m], "__file__", None) is None]
    assert deepctr.__version__ == "0.9.4"
    for path, names in {
        "deepctr.feature_column": ["SparseFeat", "VarLenSparseFeat", "DenseFeat", "get_feature_names", "build_input_features"],
        "deepctr.inputs": ["create_embedding_matrix", "embedding_lookup", "varlen_embedding_lookup", "get_varlen_pooling_list", "get_dense_input"],
        "deepctr.models": ["DeepFM", "DCN", "xDeepFM", "DIN", "WDL", "FNN", "AFM", "PNN", "NFM", "DCNMix"],
        "deepctr.models.deepfm": ["DeepFM"], "deepctr.models.dcn": ["DCN"], "deepctr.models.xdeepfm": ["xDeepFM"],
        "deepctr.models.sequence": ["DIN"], "deepctr.models.sequence.din": ["DIN"],
        "deepctr.layers": ["FM", "CrossNet", "CIN", "AFMLayer", "InnerProductLayer", "SequencePoolingLayer", "WeightedSequenceLayer",
                           "AttentionSequencePoolingLayer", "Hash", "Linear", "DNN", "PredictionLayer", "custom_objects"],
        "deepctr.layers.interaction": ["FM", "CrossNet", "CIN", "AFMLayer", "InnerProductLayer", "BiInteractionPooling", "CrossNetMix"],
        "deepctr.layers.sequence": ["SequencePoolingLayer", "WeightedSequenceLayer", "AttentionSequencePoolingLayer"],
        "deepctr.layers.core": ["DNN", "PredictionLayer", "LocalActivationUnit"],
        "deepctr.layers.activation": ["Dice"],
        "deepctr.layers.utils": ["Hash", "Linear", "Concat", "NoMask", "concat_func", "add_func", "combined_dnn_input"],
    }.items():
        mod = importlib.import_module(path)
        twin = importlib.import_module(path.replace("deepctr", "deepctr_amd", 1))
        assert mod is twin, path
        for n in names:
            assert hasattr(mod, n), "%s.%s" % (path, n)
    from deepctr.models import DeepFM
    from deepctr.models.deepfm import DeepFM as D2
    assert DeepFM is D2 is deepctr_amd.models.DeepFM
    assert deepctr.layers.custom_objects["FM"] is deepctr_amd.layers.FM


def test_example_preprocessing_reproduces_the_reference_code_fixture():
    """examples/run_classification_criteo.py:10-41 (pandas + sklearn preprocessing, feature columns through the facade)
    on the committed 200-row data file gives exactly the feed of the fixture the reference's own code produced."""
    import numpy as np
    from tests.test_gpu_facade import criteo_example_inputs
    from tests.util import load_golden
    data, cols, names = criteo_example_inputs()
    g = load_golden("model_deepfm_criteo_sample")
    assert names == ["C%d" % i for i in range(1, 27)] + ["I%d" % i for i in range(1, 14)]
    for n in names:
        ref = g["feed/" + n]
        got = data[n].values.astype(ref.dtype)
        assert np.array_equal(got, ref), n
    assert [c.vocabulary_size for c in cols[:26]] == [int(g["feed/C%d" % i].max()) + 1 for i in range(1, 27)]
