"""MI355X-native wavefront path planner behind mbf_mesh_core::MeshPlanner::makePlan.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C-ABI of
include/mnav.h), ``capi`` (ctypes binding of that C-ABI), ``planner`` (host-side mirror
of the reference plugin interface) and ``meshgen`` (synthetic inputs).
"""
__version__ = "0.1.0"
