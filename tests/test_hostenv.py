"""Host environment helper (fetalsyngen_amd/hostenv.py): CPU share of the process and the torch thread cap."""
import os

import torch

from fetalsyngen_amd import hostenv


def test_quota_line_parsing():
    assert hostenv._quota_cpus("1600000 100000") == 16
    assert hostenv._quota_cpus("150000 100000") == 2  # a fractional share rounds up
    assert hostenv._quota_cpus("max 100000") is None
    assert hostenv._quota_cpus("") is None
    assert hostenv._quota_cpus("-1 100000") is None  # cgroup v1 spells "no limit" as -1
    assert hostenv._quota_cpus("abc def") is None


def test_cpu_share_is_within_the_machine():
    n = hostenv.cpu_share()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_cap_only_lowers():
    before = torch.get_num_threads()
    try:
        assert hostenv.cap_host_threads(before + 5) == before  # never raised
        assert torch.get_num_threads() == before
        if before > 1:
            assert hostenv.cap_host_threads(1) == 1
            assert torch.get_num_threads() == 1
    finally:
        torch.set_num_threads(before)


def test_import_left_torch_within_the_share():
    # the package import applies the cap (unless FSG_KEEP_TORCH_THREADS is set)
    if not os.environ.get("FSG_KEEP_TORCH_THREADS"):
        assert torch.get_num_threads() <= max(hostenv.cpu_share(), 1)
