"""models.gc_arm — GC-ARM on MI355X (drop-in for the reference's models/gc_arm.py: same class names, constructor order
and state_dict keys; inference on armnet_gc_fused_fwd_f32)."""
from armnet_hip.siblings import GC_ARMModel, GC_SparseAttLayer  # noqa: F401

__all__ = ["GC_ARMModel", "GC_SparseAttLayer"]
