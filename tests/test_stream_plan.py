"""HipLM._plan_512 (how a batch of stateful streams is cut into groups the width-512 scans take) as plain host logic: no GPU."""
import pytest

from ocrd_keraslm_amd.lib.engine import HipLM


class _Self(object):
    pad_streams = True
    plan_override = None


@pytest.mark.parametrize("B,limit,want", [
    (300, 3072, [(300, 300)]), (512, 3072, [(512, 512)]),
    (520, 3072, [(520, 1024)]), (1000, 3072, [(1000, 1024)]), (1280, 3072, [(1280, 1536)]),
    (2560, 3072, [(2560, 2560)]), (3000, 3072, [(3000, 3072)]), (3072, 3072, [(3072, 3072)]),
    (3073, 3072, [(2560, 2560), (513, 1024)]),                       # (never a single block of 512 at the end)
    (3584, 3072, [(2560, 2560), (1024, 1024)]), (4096, 3072, [(3072, 3072), (1024, 1024)]),
    (6144, 3072, [(3072, 3072), (3072, 3072)]),
    (10000, 3072, [(3072, 3072), (3072, 3072), (3072, 3072), (784, 1024)]),
    (4096, 16383, [(3072, 3072), (1024, 1024)]),                     # (short windows: still at most six row blocks per workgroup)
    (3000, 2047, [(1536, 1536), (1464, 1536)]),                      # (long windows: fewer streams per launch sequence)
])
def test_plan(B, limit, want):
    plan = HipLM._plan_512(_Self(), B, limit)
    assert plan == want
    assert sum(n for n, _ in plan) == B
    assert all(n <= run <= max(limit, 512) and (run % 512 == 0 or run == n) for n, run in plan)


def test_windows_too_long_for_1024_streams():
    """at T >= ~1024 fewer than 1024 streams fit the 32-bit offsets of a launch sequence -- no second-generation
    count applies, and the batch must still come back in groups the kernels address, not as one oversized group"""
    plan = HipLM._plan_512(_Self(), 2000, 700)
    assert sum(n for n, _ in plan) == 2000
    assert all(n == run and n <= 700 for n, run in plan), plan
    plan = HipLM._plan_512(_Self(), 2000, 1023)
    assert sum(n for n, _ in plan) == 2000 and all(run <= 1023 for _, run in plan), plan


def test_no_padding_when_switched_off():
    s = _Self()
    s.pad_streams = False
    assert HipLM._plan_512(s, 3000, 3072) == [(3000, 3000)]
    assert HipLM._plan_512(s, 5000, 3072) == [(3072, 3072), (1928, 1928)]
