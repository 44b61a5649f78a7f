"""Synthetic level-of-detail trees in the buffer layout TensorTree keeps (LoG/model/tensor_tree.py:57-90): see
log_amd.scenes.synth_tree (shared with bench.py's C3 leg)."""
from log_amd.scenes import synth_tree  # noqa: F401
