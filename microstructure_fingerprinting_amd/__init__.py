"""MI355X-native per-voxel fingerprint matcher behind the API of
rensonnetg/microstructure_fingerprinting (``MFModel.fit`` / ``mf_utils``).

    import microstructure_fingerprinting_amd as mf
    model = mf.MFModel(dictionary)            # str path to a .mat file or a dict
    fit = model.fit(data, mask, numfasc, peaks=..., pgse_scheme=...)
    mf.mf_utils.solve_exhaustive_posweights(A, y, dicsizes)
"""
__version__ = "0.1.0"

from . import mf_utils  # noqa: E402,F401
from .mf import MFModel, MFModelFit, cleanup_2fascicles  # noqa: E402,F401
