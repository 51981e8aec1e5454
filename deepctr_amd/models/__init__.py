"""In-scope model constructors under the reference's names (deepctr/models/__init__.py:1-27 exports 27;
BASELINE north_star scopes this build to the four below)."""
from .dcn import DCN
from .deepfm import DeepFM
from .sequence import DIN
from .xdeepfm import xDeepFM
