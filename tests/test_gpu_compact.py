"""The compaction primitive by itself (vk_compact_offsets = util.cuh:52-140 PrefixSum<N>),
against the reference's own known-answer test, tests/util_test.cu:63-98: element i holds
i % 7 items, sizes 1 ... 1025; every element writes its index into its slots; the total must
be the sum, and walking the output every index must appear in one run of exactly i % 7
entries. Plus what the reference's atomics cannot promise: the ranges are in input order."""
import numpy as np
import pytest

from test_gpu_parity import api, sync  # noqa: F401

pytestmark = pytest.mark.gpu


def compact(api, counts):
    import torch
    n = len(counts)
    d_counts = torch.from_numpy(np.asarray(counts, dtype=np.int32)).cuda()
    offsets = torch.full((max(n, 1),), -9, dtype=torch.int32, device="cuda")
    total = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.empty(max(int(api.lib().vk_compact_workspace_bytes(n)), 4), dtype=torch.uint8, device="cuda")
    api.check(api.lib().vk_compact_offsets(api._ptr(d_counts), n, api._ptr(offsets), api._ptr(total), api._ptr(ws),
                                           api.stream()), "vk_compact_offsets")
    sync()
    return offsets.cpu().numpy()[:n], int(total.cpu()[0]), (d_counts, offsets, total, ws)


@pytest.mark.parametrize("count", [1, 2, 31, 32, 33, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 4097, 1 << 20])
def test_prefix_sum_known_answer(api, count):
    counts = np.arange(count, dtype=np.int32) % 7                      # util_test.cu:27-30
    offsets, total, _ = compact(api, counts)
    assert total == int(counts.sum())                                   # :78
    # the kernel of util_test.cu:32-35: output[offset + i] = index
    output = np.full(total, -1, dtype=np.int64)
    for i in np.nonzero(counts)[0]:
        assert offsets[i] >= 0
        output[offsets[i]:offsets[i] + counts[i]] = i
    assert np.all(offsets[counts == 0] == -1)                           # util.cuh:93-94
    # :82-96: runs of equal indices, each exactly as long as its count
    expected = counts.copy()
    prev = None
    for index in output:
        assert index >= 0
        if prev is not None and prev != index:
            assert expected[prev] == 0
        expected[index] -= 1
        prev = index
    assert np.all(expected == 0)
    # stronger than the reference: input order (exclusive prefix sum)
    want = np.concatenate([[0], np.cumsum(counts)[:-1]])
    assert np.array_equal(offsets[counts > 0], want[counts > 0])


def test_total_accumulates_and_edge_cases(api):
    import torch
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 5, 5000).astype(np.int32)
    counts[rng.random(5000) < 0.6] = 0
    offsets, total, (d_counts, d_offsets, d_total, ws) = compact(api, counts)
    assert total == counts.sum()
    # a second call adds to the same total, its ranges start where the first ended (the reference's `total`)
    api.check(api.lib().vk_compact_offsets(api._ptr(d_counts), len(counts), api._ptr(d_offsets), api._ptr(d_total),
                                           api._ptr(ws), api.stream()), "vk_compact_offsets")
    sync()
    assert int(d_total.cpu()[0]) == 2 * total
    second = d_offsets.cpu().numpy()
    assert np.array_equal(second[counts > 0], offsets[counts > 0] + total)
    zeros, t0, _ = compact(api, np.zeros(777, np.int32))
    assert t0 == 0 and np.all(zeros == -1)
    lib = api.lib()
    assert lib.vk_compact_offsets(None, 0, None, None, None, None) == 0          # nothing to do
    assert lib.vk_compact_offsets(None, 5, None, None, None, None) == -1
    assert lib.vk_compact_workspace_bytes(1025) == 8 and lib.vk_compact_workspace_bytes(0) == 0
