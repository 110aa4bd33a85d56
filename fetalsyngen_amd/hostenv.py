"""Host-side environment of a generator process: how many CPUs it may use, and torch's CPU thread count against that.

Why this exists: the path's host work is small (draw plans, slice transforms, a few compacted arrays), but PyTorch sizes its
intra-op pool by the machine (128 threads on an MI355X host) even when the process is confined to a share of it (16 CPUs per
GPU in a container).  One parallel CPU op then wakes 128 spinning workers inside a 16-CPU quota, the cgroup is throttled for
the rest of the scheduler period, and an SR-artifact stage that takes 5 ms takes 95 ms in four repetitions out of ten
(profiles/r03_j_host_threads.txt).  `cap_host_threads()` lowers torch's count to the CPUs the process really has; it never
raises it.  Called once when the package is imported (set FSG_KEEP_TORCH_THREADS=1 to leave torch alone).
"""
from __future__ import annotations

import math
import os


def _quota_cpus(cpu_max_text: str):
    """CPUs granted by a cgroup-v2 `cpu.max` line ("<quota> <period>" or "max <period>"); None when unlimited / unreadable."""
    parts = cpu_max_text.split()
    if len(parts) != 2 or parts[0] == "max":
        return None
    try:
        quota, period = int(parts[0]), int(parts[1])
    except ValueError:
        return None
    if quota <= 0 or period <= 0:
        return None
    return max(1, math.ceil(quota / period))


def _cgroup_cpus():
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2
            return _quota_cpus(f.read().strip())
    except OSError:
        pass
    try:  # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = f.read().strip()
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = f.read().strip()
        return _quota_cpus(f"{quota} {period}")
    except OSError:
        return None


def cpu_share() -> int:
    """CPUs this process may run on: scheduler affinity, cut by the cgroup's CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = _cgroup_cpus()
    if q is not None:
        n = min(n, q)
    return max(1, n)


def cap_host_threads(limit: int | None = None) -> int:
    """Lower torch's intra-op CPU thread count to `limit` (default: cpu_share()) if it is above it; returns the count in force."""
    import torch

    limit = cpu_share() if limit is None else max(1, int(limit))
    cur = torch.get_num_threads()
    if cur > limit:
        torch.set_num_threads(limit)
        cur = limit
    return cur
