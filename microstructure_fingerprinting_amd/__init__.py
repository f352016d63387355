"""MI355X-native per-voxel fingerprint matcher behind the API of
rensonnetg/microstructure_fingerprinting (MFModel.fit / mf_utils)."""
__version__ = "0.1.0"
