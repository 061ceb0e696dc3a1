"""raht-3dgs-codec_amd -- MI355X-native RAHT attribute codec hot path (forward / inverse RAHT, its
plan, quantize / reorder and the on-device voxelizer) behind the reference's ``raht_fn`` operator
table.  See DESIGN.md and include/raht.h.  Import name: ``raht_3dgs_codec_amd``."""
from ._lib import RahtError, SO_PATH, build  # noqa: F401
from .ops import (RAHT2_optimized, RAHT_param_reorder_fast, RahtPlan, get_morton_code,  # noqa: F401
                  inverse_RAHT_optimized, plan_of, raht_fn, sort_keys, voxelize_pc_batched, voxelize_plan)

from . import merge, pipeline, ply_io, rlgr, sharded, synth  # noqa: F401,E402

__all__ = ["raht_fn", "RAHT2_optimized", "inverse_RAHT_optimized", "RAHT_param_reorder_fast", "RahtPlan",
           "plan_of", "voxelize_pc_batched", "voxelize_plan", "get_morton_code", "sort_keys", "RahtError", "build"]
