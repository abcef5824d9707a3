"""Host-side model of k_scan_rank (fast-livo2_amd/csrc/frame_kernels.hpp): the position of a point in the stable order by Morton key is COUNTED,
position(i) = #{j : (key_j, j) < (key_i, i)}, by blocks of 64 points x 16 waves, each wave over one sixteenth of the (padded) keys, with ONE compare per pair outside the
stretch of keys that surrounds the block's own points (threshold key_i + 1 before it, key_i behind it) and the index tie-break only inside it.  This test re-states exactly
that partition in numpy — the ranges, the roundings to 16, the thresholds — and holds it against numpy's stable argsort for sizes around every boundary and for key sets
with many ties; the device kernel itself is compared with the library's stable radix sort in tests/test_frame_ingest_gpu.py and with std::stable_sort in
tools/sort_probe.hip.  (What the order is for: voxel_map.cpp:349-360 / LIVMapper.cpp:351-352 hand a scan to StateEstimation; the order only fixes the summation order of
the 29 partial sums, so it must be THE stable order, not just a good one, for results to be independent of the path.)"""
import numpy as np
import pytest

WAVE, RANK_WAVES = 64, 16


def model_positions(keys):
    n = len(keys)
    padded = (n + 15) & ~15
    k = np.full(padded, 0xFFFFFFFF, np.uint64)
    k[:n] = keys
    P = (((padded + RANK_WAVES - 1) // RANK_WAVES) + 15) & ~15
    pos = np.zeros(n, np.int64)
    for i0 in range(0, n, WAVE):
        idx = np.arange(i0, min(i0 + WAVE, n))
        ki = k[idx]
        cnt = np.zeros(len(idx), np.int64)
        for w in range(RANK_WAVES):
            jb = min(w * P, padded); je = min(jb + P, padded)
            lo_end = min(max(i0 & ~15, jb), je)
            hi_beg = min(max((i0 + WAVE + 15) & ~15, jb), je)
            assert jb % 16 == 0 and je % 16 == 0 and lo_end % 16 == 0 and hi_beg % 16 == 0 and jb <= lo_end <= hi_beg <= je
            assert lo_end <= i0 or lo_end == jb                              # every j of the first stretch precedes every point of the block
            assert hi_beg >= i0 + WAVE or hi_beg == je
            cnt += (k[jb:lo_end][None, :] < (ki + 1)[:, None]).sum(1)       # ties before the block count
            mid = np.arange(lo_end, hi_beg)
            km = k[lo_end:hi_beg]
            cnt += ((km[None, :] < ki[:, None]) | ((km[None, :] == ki[:, None]) & (mid[None, :] < idx[:, None]))).sum(1)
            cnt += (k[hi_beg:je][None, :] < ki[:, None]).sum(1)              # ties behind the block do not
        pos[idx] = cnt
    return pos


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 4095, 4096, 4097, 10761, 16383, 16384])
@pytest.mark.parametrize("key_bits", [30, 6, 0])
def test_counted_positions_are_the_stable_order(n, key_bits):
    rng = np.random.default_rng(n * 31 + key_bits)
    keys = rng.integers(0, 1 << key_bits, size=n, dtype=np.uint64) if key_bits else np.zeros(n, np.uint64)       # 6 bits: ~n/64 ties per key; 0 bits: all equal
    if n > 4097 and key_bits == 30:
        keys[rng.integers(0, n, size=n // 3)] = keys[0]                       # a heavy tie class next to unique keys
    pos = model_positions(keys)
    order = np.argsort(keys, kind="stable")
    want = np.empty(n, np.int64); want[order] = np.arange(n)
    assert np.array_equal(pos, want)
    assert np.array_equal(np.sort(pos), np.arange(n))                        # a permutation: every output slot written exactly once


FRAME_COPY_BLOCKS, FRAME_THREADS = 64, 256


def model_copy_counts(nbytes):
    """how often every byte of a segment is written by the copy blocks of k_frame_ingest (frame_kernels.hpp): 16-byte units, four per thread and trip while four strides
    fit, then one per trip, then the < 16 tail bytes by the first threads of block 0"""
    units, tail = nbytes >> 4, nbytes & 15
    stride = FRAME_COPY_BLOCKS * FRAME_THREADS
    hits = np.zeros(nbytes, np.int64)
    u0 = np.arange(stride)                                                  # one entry per (block, thread)
    u = u0.copy()
    while True:
        go = u + 3 * stride < units
        if not go.any():
            break
        for q in range(4):
            for x in (u[go] + q * stride):
                hits[x * 16:(x + 1) * 16] += 1
        u = np.where(go, u + 4 * stride, u)
    while True:
        go = u < units
        if not go.any():
            break
        for x in u[go]:
            hits[x * 16:(x + 1) * 16] += 1
        u = np.where(go, u + stride, u)
    for t in range(tail):
        hits[(units << 4) + t] += 1
    return hits


@pytest.mark.parametrize("nbytes", [0, 1, 15, 16, 17, 6280, 4096 * 16, 4096 * 16 + 4, 64 * 256 * 16 - 16, 64 * 256 * 16, 64 * 256 * 16 * 4 + 12, 64 * 256 * 16 * 5 + 16 * 77 + 3, 358400, 327680])
def test_every_byte_of_a_segment_is_copied_exactly_once(nbytes):
    hits = model_copy_counts(nbytes)
    assert hits.shape == (nbytes,) and (hits == 1).all()
