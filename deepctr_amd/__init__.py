"""deepctr_amd — MI355X-native forward path for DeepCTR's embedding lookup + feature interaction
(see DESIGN.md).  No import-time side effects (the reference starts an HTTP version check in
deepctr/__init__.py:1-4; that is deliberately not reproduced)."""
__version__ = "0.1.0"
