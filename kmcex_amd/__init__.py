"""kmcex_amd -- MI355X-native KModel insert/query hot path (drop-in for lzhLab/kmcEx's kmodel.hpp path).

Layout: csrc/ (gfx950 HIP kernels + the C ABI of include/kmx.h), api.py (Python mirror of the reference's
KModel interface over that ABI), synth.py / kmcdb.py (synthetic streams and a KMC1 writer: plumbing).
"""
from .api import KModel, KmxError, get_model, load_library, device_count  # noqa: F401
