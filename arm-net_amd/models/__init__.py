"""`models` package of the MI355X build.

Only the ARM-Net hot-path modules live here (armnet, armnet_1h, layers).  When this directory is put
AHEAD of a checkout of the reference on sys.path, the package path is extended with the reference's own
`models/` directory, so `models.model_utils.create_model` and the baseline models keep importing from the
reference while `models.armnet`, `models.armnet_1h` and `models.layers.{Embedding,MLP}` resolve here.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
