"""tools/ only: the probes and sweeps of rounds 1-2 pass plan-time knobs as FLEX_* environment variables.  The library reads
none of them since ABI 3 (flex_plan_tuning); importing this module makes every flex_amd.Plan() created by a TOOL pick the
variables up at that moment and hand them over in the plan descriptor, so the old command lines keep working:

    FLEX_LANES=8 FLEX_WAVE_NNZ=384 python tools/sweep.py ...
"""
import os

import flex_amd
from flex_amd import binding

ENV_TO_KNOB = {
    "FLEX_LANES": "lanes_per_nz", "FLEX_WAVE_NNZ": "chunk_records", "FLEX_LONG_ROW": "long_row", "FLEX_PIECE": "piece_records",
    "FLEX_ROW_COST": "row_cost", "FLEX_XCD_REMAP": "xcd_slices", "FLEX_XCD_BALANCE": "xcd_balance", "FLEX_XCD_STRETCH": "xcd_stretch", "FLEX_CHUNK_COST": "chunk_cost",
    "FLEX_TASK_COST": "task_cost", "FLEX_FUSED_FIXUP": "split_rows", "FLEX_REC_NT": "rec_nt", "FLEX_U": "unroll", "FLEX_2D": "two_d",
    "FLEX_PANEL_KB": "panel_kb", "FLEX_SEG_MIN": "seg_min", "FLEX_MFMA": "mfma", "FLEX_MFMA_FILL": "mfma_fill_pct",
    "FLEX_LDS_EXTRA": "lds_extra", "FLEX_HOST_THREADS": "host_threads", "FLEX_CLUSTER_BATCH": "cluster_batch",
    "FLEX_CLUSTER_NO_REFINE": "cluster_no_refine", "FLEX_CLUSTER_STRETCH": "cluster_stretch", "FLEX_CLUSTER_SWEEPS": "cluster_sweeps",
    "FLEX_CLUSTER_STRIDE": "cluster_stride", "FLEX_BLOCKS": "blocks", "FLEX_BLOCK_ROUNDS": "block_rounds", "FLEX_BLOCK_PANEL_ROWS": "block_panel_rows",
    "FLEX_BLOCK_THR": "block_thr", "FLEX_BLOCK_CAP": "block_cap", "FLEX_TILE_GROUP": "tile_group", "FLEX_FAR_FIRST": "far_first",
    "FLEX_BUNDLE": "bundle", "FLEX_BUNDLE_LEN": "bundle_len",
}


def env_tuning() -> dict:
    t = {}
    for env, knob in ENV_TO_KNOB.items():
        v = os.environ.get(env)
        if v is None or v == "":
            continue
        v = int(v)
        if env == "FLEX_2D" and v == 2:  # "2 = off" in the old convention; 0 = rule = off
            v = 0
        t[knob] = v
    return t


def sync():
    binding.DEFAULT_TUNING.clear()
    binding.DEFAULT_TUNING.update(env_tuning())


_plan_init = binding.Plan.__init__


def _init_with_env(self, *a, **kw):
    sync()
    _plan_init(self, *a, **kw)


binding.Plan.__init__ = _init_with_env
flex_amd.Plan = binding.Plan
