"""semantic_slam_mapping_amd -- MI355X (gfx950) per-frame semantic-mapping front end.

The product is the C-ABI library ``libssm_hip.so`` (include/ssm_hip.h, sources in csrc/) and the C++ host classes in
include/ssm/.  This package is the Python mirror of that boundary used by tests and bench.py: thin ctypes calls,
numpy only for host buffers.  It has no CPU fallback -- a missing library or a missing GPU raises.
"""
from .api import Context, Tracker, GlibcRand, SsmError, default_config, KEYPOINT_DTYPE, DMATCH_DTYPE, POINT_DTYPE, VOXEL_DTYPE, PMATCH_DTYPE  # noqa: F401
from ._lib import LIB_PATH, SYMBOLS, load  # noqa: F401
