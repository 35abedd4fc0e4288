"""Drop-in for the hot-path part of the reference's models/layers.py.

Only the two layers ARM-Net uses are implemented here: ``Embedding``
(layers.py:8-21, HIP gather*value) and ``MLP`` (layers.py:68-88, the prediction head that bounds the
fused kernel).  If a checkout of the reference follows on sys.path, every OTHER public name of its
models/layers.py (Linear, FactorizationMachine, ... used by the baseline models) is re-exported from
there at import time, so the reference's model factory keeps working; nothing is copied.
"""
import importlib.util
import os

from armnet_hip.modules import HipEmbedding as Embedding  # noqa: F401
from armnet_hip.modules import _MLP as MLP  # noqa: F401

__all__ = ["Embedding", "MLP"]


def _reexport_reference_layers():
    import models
    here = os.path.dirname(os.path.abspath(__file__))
    for d in list(models.__path__):
        cand = os.path.join(d, "layers.py")
        if os.path.abspath(d) != here and os.path.isfile(cand):
            spec = importlib.util.spec_from_file_location("models._reference_layers", cand)
            ref = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(ref)
            for k, v in vars(ref).items():
                if not k.startswith("_") and k not in ("Embedding", "MLP") and k not in globals():
                    globals()[k] = v
            return


_reexport_reference_layers()
