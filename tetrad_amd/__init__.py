"""tetrad_amd -- MI355X-native quartet-invariant engine behind tetrad's worker API.

Only the per-quartet hot path of eaton-lab/tetrad lives here:
HIP kernels + C ABI in ``csrc/``, the ctypes binding, and a host-side mirror of
the reference worker interface.  Importing the package does not load the HIP
library; the first compute call does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"
