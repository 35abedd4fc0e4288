"""Drop-in for the hot-path part of the reference's models/layers.py.

Only the two layers ARM-Net uses are provided: ``Embedding``
(layers.py:8-21, HIP gather*value) and ``MLP`` (layers.py:68-88, the prediction head that
bounds the fused kernel).  The baseline-model helpers of that file are out of scope.
"""
from armnet_hip.modules import HipEmbedding as Embedding  # noqa: F401
from armnet_hip.modules import _MLP as MLP  # noqa: F401

__all__ = ["Embedding", "MLP"]
