"""``deepctr`` — the reference's import name over the MI355X-native build (``deepctr_amd``).

BASELINE north_star: the path "drops in under ``deepctr.models.{DeepFM,DCN,xDeepFM,DIN}``"; SURVEY.md §8(b) names the module
paths a user's code imports: ``deepctr.feature_column``, ``deepctr.inputs``, ``deepctr.models`` (and
``deepctr.models.<name>``, ``deepctr.models.sequence.din``), ``deepctr.layers`` (and ``deepctr.layers.{core,interaction,
sequence,activation,utils}``).  Every one of those modules IS the ``deepctr_amd`` module of the same name (registered under
both names in ``sys.modules``), so classes are identical objects whichever way they are imported and the reference's example
scripts (``examples/run_classification_criteo.py``, ``run_din.py`` ...) run unmodified.

Differences from the reference package, on purpose: importing it performs no HTTP version check
(reference ``deepctr/__init__.py:1-4`` -> ``utils.check_version`` starts a thread that queries PyPI), and only the §8
scope exists: the other 21 model constructors, the Estimator API and ``deepctr.contrib`` raise ``ImportError`` /
``AttributeError`` by absence.
"""
import importlib
import sys

__version__ = "0.9.4"           # API level of the reference this build mirrors (reference deepctr/__init__.py:3)

_ALIASES = (
    "feature_column", "inputs",
    "layers", "layers.activation", "layers.core", "layers.interaction", "layers.sequence", "layers.utils",
    "models", "models.afm", "models.dcn", "models.dcnmix", "models.deepfm", "models.fnn", "models.nfm", "models.pnn",
    "models.wdl", "models.xdeepfm", "models.sequence", "models.sequence.din",
)
for _name in _ALIASES:
    _mod = importlib.import_module("deepctr_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    _parent, _, _leaf = _name.rpartition(".")
    if not _parent:
        setattr(sys.modules[__name__], _leaf, _mod)
del importlib, _name, _mod, _parent, _leaf
