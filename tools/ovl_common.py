"""Shared by tools/ovl_debug*.py / ovl_bench.py: the side-stream feature generator (slr_sfs_amd.pipeline, overlap=True)."""
from slr_sfs_amd.pipeline import _features_ahead


def features_ahead_overlap(clip, frames):
    return _features_ahead(clip, frames, overlap=True)
