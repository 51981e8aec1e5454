"""In-scope model constructors under the reference's names (deepctr/models/__init__.py:1-27 exports 27;
BASELINE north_star scopes this build to DeepFM, DCN, xDeepFM and DIN; WDL and FNN are SURVEY §8(f) rank-4 siblings
that are DeepFM's graph minus terms)."""
from .dcn import DCN
from .deepfm import DeepFM
from .fnn import FNN
from .sequence import DIN
from .wdl import WDL
from .xdeepfm import xDeepFM
