"""Near-duplicate grid filter of result trajectories -- SURVEY.md section 8(f2).

``apply_trajectory_grid_filter(trajectories, bin_width, max_dt)`` has the reference's signature and
result (src/kbmod/filters/clustering_grid.py:152-175: the best trajectory of every (start bin, end
bin) key, keys in order of first occurrence) but runs ``kb_grid_filter`` on the device instead of a
Python dictionary loop; ``TrajectoryClusterGrid`` -- the online, one-at-a-time structure -- keeps the
reference's interface on the host.  The batch function raises ``RuntimeError`` without a GPU.
"""

import numpy as np

from . import search as _search


class TrajectoryClusterGrid:
    """A spatial hash of trajectory results (clustering_grid.py:13-149)."""

    def __init__(self, bin_width=10, max_time=1.0):
        if bin_width < 1 or not np.isfinite(bin_width):
            raise ValueError(f"Bin width must be at least 1. Got {bin_width}.")
        self.bin_width = bin_width
        if max_time < 0 or not np.isfinite(max_time):
            raise ValueError(f"Max time must be >= 0. Got {max_time}.")
        self.max_time = max_time
        self.table = {}
        self.count = {}
        self.idx_table = {}
        self.total_count = 0

    def __len__(self):
        return len(self.table)

    def _key(self, trj):
        return (int(trj.x / self.bin_width), int(trj.y / self.bin_width),
                int((trj.x + self.max_time * trj.vx) / self.bin_width),
                int((trj.y + self.max_time * trj.vy) / self.bin_width))

    def add_trajectory(self, trj, idx=None):
        if idx is None:
            idx = self.total_count
        key = self._key(trj)
        if key not in self.table:
            self.table[key] = trj
            self.count[key] = 1
            self.idx_table[key] = idx
        else:
            if trj.lh > self.table[key].lh:
                self.table[key] = trj
                self.idx_table[key] = idx
            self.count[key] += 1
        self.total_count += 1

    def add_trajectory_list(self, trj_list):
        for idx, trj in enumerate(trj_list):
            self.add_trajectory(trj, idx=idx)

    def get_trajectories(self):
        return list(self.table.values())

    def get_indices(self):
        return list(self.idx_table.values())


def apply_trajectory_grid_filter(trajectories, bin_width, max_dt):
    """(surviving trajectories, their indices), computed on the device."""
    if bin_width < 1 or not np.isfinite(bin_width):
        raise ValueError(f"Bin width must be at least 1. Got {bin_width}.")
    if max_dt < 0 or not np.isfinite(max_dt):
        raise ValueError(f"Max time must be >= 0. Got {max_dt}.")
    trajectories = list(trajectories)
    indices = [int(i) for i in _search.grid_filter_indices(trajectories, float(bin_width), float(max_dt))]
    return [trajectories[i] for i in indices], indices
