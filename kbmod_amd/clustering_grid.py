"""Near-duplicate grid filter of result trajectories -- SURVEY.md section 8(f2).

Two entry points with the reference's names and behaviour (src/kbmod/filters/clustering_grid.py):

* ``apply_trajectory_grid_filter(trajectories, bin_width, max_dt)`` (lines 152-175, called from
  run_search.py:294-301): trajectories that share a (start bin, end bin at ``max_dt``) key are duplicates, the most
  likely one of each key survives (the earliest of equals), keys come out in order of first occurrence.  The
  reference walks a Python dictionary; here ``kb_grid_filter`` sorts and scans in HBM.  Raises ``RuntimeError``
  without a GPU.
* ``TrajectoryClusterGrid`` (lines 13-149): the online form -- one trajectory (or one list) at a time, with the
  ``table`` / ``count`` / ``idx_table`` dictionaries and ``total_count`` that callers such as
  filters/clustering_filters.py read.  Host bookkeeping by nature (a dictionary that is queried between
  insertions); kept so that code written against the reference finds it.
"""

import math

from . import search as _search


def _check_grid_arguments(bin_width, max_time):
    if not (math.isfinite(bin_width) and bin_width >= 1):
        raise ValueError(f"Bin width must be at least 1. Got {bin_width}.")
    if not (math.isfinite(max_time) and max_time >= 0):
        raise ValueError(f"Max time must be >= 0. Got {max_time}.")


class TrajectoryClusterGrid:
    """Spatial hash of trajectories keyed by (start bin x, start bin y, end bin x, end bin y); every key keeps its
    most likely member (a later one only if strictly more likely), the member's index and the number of
    trajectories that fell on the key."""

    def __init__(self, bin_width=10, max_time=1.0):
        _check_grid_arguments(bin_width, max_time)
        self.bin_width = bin_width
        self.max_time = max_time
        self.table = {}      # key -> best trajectory
        self.count = {}      # key -> members seen
        self.idx_table = {}  # key -> index of the best trajectory
        self.total_count = 0

    def __len__(self):
        return len(self.table)

    def bin_key(self, trj):
        """Bins are truncated quotients (``int()`` rounds towards zero, as the reference's keys do)."""
        w, dt = self.bin_width, self.max_time
        return (int(trj.x / w), int(trj.y / w), int((trj.x + dt * trj.vx) / w), int((trj.y + dt * trj.vy) / w))

    def _offer(self, key, trj, idx):
        holder = self.table.get(key)
        if holder is None or trj.lh > holder.lh:
            self.table[key] = trj
            self.idx_table[key] = idx
        self.count[key] = self.count.get(key, 0) + 1

    def add_trajectory(self, trj, idx=None):
        """``idx`` defaults to the number of trajectories added so far."""
        self._offer(self.bin_key(trj), trj, self.total_count if idx is None else idx)
        self.total_count += 1

    def add_trajectory_list(self, trj_list):
        """Indices are positions within ``trj_list`` (the reference's convention for the list form)."""
        for position, trj in enumerate(trj_list):
            self._offer(self.bin_key(trj), trj, position)
        self.total_count += len(trj_list)

    def get_trajectories(self):
        return list(self.table.values())

    def get_indices(self):
        return list(self.idx_table.values())


def apply_trajectory_grid_filter(trajectories, bin_width, max_dt):
    """(surviving trajectories, their indices into the input), computed on the device."""
    _check_grid_arguments(bin_width, max_dt)
    rows = list(trajectories)
    keep = [int(i) for i in _search.grid_filter_indices(rows, float(bin_width), float(max_dt))]
    return [rows[i] for i in keep], keep
