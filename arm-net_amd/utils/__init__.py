"""`utils` package of the MI355X build (entmax only); extended with a reference checkout's `utils/` when one
follows on sys.path (train.py imports utils.utils from there)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
