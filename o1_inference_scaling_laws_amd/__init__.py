"""o1_inference_scaling_laws_amd -- MI355X-native self-consistency aggregation engine.

A drop-in for ONE hot path of hughbzhang/o1_inference_scaling_laws: the majority vote
(/root/reference/o1.py:181-213), the per-budget reduction (o1.py:229-247) and the token-budget
bucketing (o1.py:266-283, 297-308) that feeds helpers/plot_helpers.py.  The arithmetic runs in
hand-written HIP kernels for gfx950 behind the C ABI of ``include/scvote.h``; this package is the
Python host side (ctypes binding + the mirror of the reference's function interface).

There is NO CPU fallback: without ``csrc/libscvote.so`` and a HIP device the engine raises.
"""
from .scoring import accuracy_from_tie_classes, avg_tokens_used, pass_at_k  # noqa: F401

__all__ = ["accuracy_from_tie_classes", "avg_tokens_used", "pass_at_k"]
__version__ = "0.1.0"
