"""tetrad_amd -- MI355X-native quartet-invariant engine behind tetrad's worker API.

Only the per-quartet hot path of eaton-lab/tetrad and the callers / consumers either side of it live here
(SURVEY.md section 8): HIP kernels + C ABI in ``csrc/``, the ctypes binding (``_lib``, ``engine``) and host-side
mirrors of the reference's interfaces -- ``resolve_quartets`` (worker), ``distributor`` (dispatch + result gather
over the GPUs of a node), ``replicates`` (bootstrap-replicate loop), ``bootstrap`` (resampler draws),
``combinations`` (quartet producers), ``qmc_format`` / ``qmc`` (wQMC text and the quartet supertree).
Importing the package does not load the HIP library; the first compute call does, and fails loudly if it is missing.
"""
__version__ = "0.2.0"
