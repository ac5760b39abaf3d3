"""rpgp_amd — MI355X-native additive randomly-projected GP kernel MVM + exact-GP solve.

The directory is named `randomly-projected-additive-gps_amd`; import it through the `rpgp_amd` shim at the
repository root.  The compute path is the HIP library in csrc/ (C-ABI: include/rpgp.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
