"""Near-duplicate grid filter of result trajectories on the device -- SURVEY.md section 8(f2).

``apply_trajectory_grid_filter(trajectories, bin_width, max_dt)`` has the reference's signature and result
(src/kbmod/filters/clustering_grid.py:152-175, called from run_search.py:294-301): trajectories that share a
(start bin, end bin at ``max_dt``) key are duplicates, the most likely one of each key survives (the earliest
of equals), keys come out in order of first occurrence.  The reference walks a Python dictionary; here
``kb_grid_filter`` sorts and scans in HBM.  Raises ``RuntimeError`` without a GPU.  (The reference's online
``TrajectoryClusterGrid`` -- insert one trajectory at a time -- is host-only bookkeeping; its behaviour is what
oracle/post_search.py's ``grid_filter_indices`` restates and the device result is tested against.)
"""

import math

from . import search as _search


def apply_trajectory_grid_filter(trajectories, bin_width, max_dt):
    """(surviving trajectories, their indices into the input), computed on the device."""
    if not (math.isfinite(bin_width) and bin_width >= 1):
        raise ValueError(f"Bin width must be at least 1. Got {bin_width}.")
    if not (math.isfinite(max_dt) and max_dt >= 0):
        raise ValueError(f"Max time must be >= 0. Got {max_dt}.")
    rows = list(trajectories)
    keep = [int(i) for i in _search.grid_filter_indices(rows, float(bin_width), float(max_dt))]
    return [rows[i] for i in keep], keep
