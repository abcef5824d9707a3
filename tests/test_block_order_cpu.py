"""The block-order counting sort of k_lidar_solve (fast-livo2_amd/csrc/lidar_kernels.hpp, lidar_block_order_wave), restated lane by lane in numpy: whatever the
lifetimes are — all equal, zeros, one outlier, a slow device whose blocks all live longer than any fixed grid — the result must be a permutation of the chunks with
non-increasing bucket index.  (The HIP code itself is exercised on the GPU by tests/test_bench_workload_gpu.py::test_block_order_...; this pins the algorithm.)"""
import numpy as np
import pytest

WAVE, SLOTS = 64, 16


def block_order(cost):
    chunks = len(cost)
    assert chunks <= WAVE * SLOTS
    cc = np.zeros((SLOTS, WAVE), np.uint32)                     # cc[u][lane] = cost[lane + 64 u], 0 beyond the scan
    valid = np.zeros((SLOTS, WAVE), bool)
    for u in range(SLOTS):
        for lane in range(WAVE):
            c = lane + WAVE * u
            if c < chunks:
                cc[u, lane], valid[u, lane] = cost[c], True
    lo = np.uint32(cc[valid].min()); hi = np.uint32(cc[valid].max())
    span = max(int(hi) - int(lo), 1)
    bk = ((np.maximum(cc, lo).astype(np.uint64) - np.uint64(lo)) * np.uint64(63)) // np.uint64(span)
    assert bk[valid].max() <= 63
    hist = np.zeros(64, np.int64)
    for u in range(SLOTS):
        for lane in range(WAVE):
            if valid[u, lane]:
                hist[bk[u, lane]] += 1
    v = hist[63 - np.arange(WAVE)]                              # lane l <-> bucket 63 - l
    fill = np.zeros(64, np.int64)
    fill[63 - np.arange(WAVE)] = np.cumsum(v) - v               # exclusive prefix over the lanes
    order = np.full(chunks, -1, np.int64)
    for u in range(SLOTS):                                      # (the device's atomics may serve the lanes in any order inside a bucket)
        for lane in np.random.default_rng(u).permutation(WAVE):
            if valid[u, lane]:
                b = bk[u, lane]
                order[fill[b]] = lane + WAVE * u
                fill[b] += 1
    return order, bk[valid].reshape(-1)


@pytest.mark.parametrize("case", ["typical", "equal", "zeros", "outlier", "slow_device", "tiny_span", "max_chunks", "huge"])
def test_order_is_a_permutation_longest_first(case):
    rng = np.random.default_rng(7)
    n = {"max_chunks": 1024, "tiny_span": 513}.get(case, 784)
    cost = {"typical": rng.normal(680, 90, n).clip(400, 1400), "equal": np.full(n, 700.0), "zeros": np.zeros(n), "outlier": np.r_[np.full(n - 1, 650.0), 90000.0],
            "slow_device": rng.normal(1500, 120, n), "tiny_span": 700 + rng.integers(0, 2, n), "max_chunks": rng.normal(680, 90, n).clip(1), "huge": rng.integers(0, 2**32 - 1, n)}[case]
    cost = np.asarray(cost).astype(np.uint32)
    order, _ = block_order(cost)
    assert sorted(order.tolist()) == list(range(n))                                        # every chunk exactly once: the launch covers the scan
    span = max(int(cost.max()) - int(cost.min()), 1)
    b = ((cost[order].astype(np.uint64) - np.uint64(cost.min())) * np.uint64(63)) // np.uint64(span)
    assert np.all(np.diff(b.astype(np.int64)) <= 0)                                        # buckets descend: longer-lived chunks start first
    if case in ("typical", "slow_device"):
        assert cost[order[:32]].mean() > cost[order[-32:]].mean() + 100
