"""Build the product nn.Modules from a golden fixture (constructor args + state_dict)."""
import numpy as np
import torch


def build_model(meta, sd=None, device=None):
    from models.armnet import ARMNetModel as MH
    from models.armnet_1h import ARMNetModel as M1
    c = meta["ctor"]
    if meta["variant"] == "1h":
        m = M1(c["nfield"], c["nfeat"], c["nemb"], c["alpha"], c["nhid"], c["d_k"], c["mlp_nlayer"],
               c["mlp_nhid"], c["dropout"], c["ensemble"], c["deep_nlayer"], c["deep_nhid"])
    else:
        m = MH(c["nfield"], c["nfeat"], c["nemb"], c["nhead"], c["alpha"], c["nhid"], c["mlp_nlayer"],
               c["mlp_nhid"], c["dropout"], c["ensemble"], c["deep_nlayer"], c["deep_nhid"])
    if sd is not None:
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    m.eval()
    if device is not None:
        m = m.to(device)
    return m
