"""mfrec_amd -- MI355X-native latent-factor training (SGD families + ALS/WRMF) behind the mfrec
recommender surface.  Host code is NumPy + ctypes over libmfx.so (include/mfx.h); there is no CPU
fallback: importing the sub-modules works anywhere, calling a kernel needs an MI355X."""
__version__ = "0.1.0"
