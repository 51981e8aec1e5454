"""In-scope layer classes under the reference's names (deepctr/layers/__init__.py:15-54 lists the full
``custom_objects`` registry; the subset below is what DeepFM / DCN / xDeepFM / DIN and the AFM / PNN
siblings need — SURVEY.md §8a)."""
from .activation import Dice
from .core import DNN, Dense, LocalActivationUnit, PredictionLayer
from .interaction import AFMLayer, BiInteractionPooling, CIN, CrossNet, CrossNetMix, FM, InnerProductLayer
from .sequence import AttentionSequencePoolingLayer, SequencePoolingLayer, WeightedSequenceLayer
from .utils import Concat, Hash, Linear, NoMask, add_func, combined_dnn_input, concat_func

custom_objects = {
    'DNN': DNN,
    'PredictionLayer': PredictionLayer,
    'FM': FM,
    'AFMLayer': AFMLayer,
    'BiInteractionPooling': BiInteractionPooling,
    'CrossNet': CrossNet,
    'CrossNetMix': CrossNetMix,
    'CIN': CIN,
    'InnerProductLayer': InnerProductLayer,
    'LocalActivationUnit': LocalActivationUnit,
    'Dice': Dice,
    'SequencePoolingLayer': SequencePoolingLayer,
    'WeightedSequenceLayer': WeightedSequenceLayer,
    'AttentionSequencePoolingLayer': AttentionSequencePoolingLayer,
    'Hash': Hash,
    'Linear': Linear,
    'Concat': Concat,
    'NoMask': NoMask,
}
