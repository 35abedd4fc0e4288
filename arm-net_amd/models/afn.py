"""models.afn — AFN on MI355X (drop-in for the reference's models/afn.py: same class name, constructor order and
state_dict keys; inference on armnet_afn_fused_fwd_f32)."""
from armnet_hip.siblings import AFNModel  # noqa: F401

__all__ = ["AFNModel"]
