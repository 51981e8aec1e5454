"""In-scope model constructors under the reference's names (deepctr/models/__init__.py:1-27 exports 27;
BASELINE north_star scopes this build to DeepFM, DCN, xDeepFM and DIN; WDL, FNN, AFM, PNN, NFM and DCNMix
are SURVEY §8(f) rank-4 siblings on the same kernels)."""
from .afm import AFM
from .dcn import DCN
from .dcnmix import DCNMix
from .deepfm import DeepFM
from .fnn import FNN
from .nfm import NFM
from .pnn import PNN
from .sequence import DIN
from .wdl import WDL
from .xdeepfm import xDeepFM
