"""float_chain_wave (fast-livo2_amd/csrc/float_chain.hpp) on the device against the serial float loop of the reference (src/vio.cpp:1554, 1634): bit for bit, for
16 / 32 / 64 lanes per chain, for the C4 shape (4 000 errors over MP_PROC_NUM = 4 threads), tie-ridden data, wide ranges, zeros, subnormals, chains of several
passes, uneven partitions, and the inputs that send it to the serial loop (negative, infinite, NaN).  The one-lane chain the short blocks keep is checked beside it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def serial_sums(e, T):
    e = np.asarray(e, np.float32)
    M = len(e); q, r = divmod(M, T)
    out = np.zeros(T, np.float32)
    with np.errstate(all="ignore"):
        for c in range(T):
            b = c * (q + 1) if c < r else c * q + r
            n = q + 1 if c < r else q
            acc = np.float32(0.0)
            for v in e[b:b + n]:
                acc = np.float32(acc + v)
            out[c] = acc
    return out


def datasets():
    rng = np.random.default_rng(20260930)
    yield "c4 shape", (300.0 * (0.05 + rng.random(4000)) * (0.05 + rng.random(4000))).astype(np.float32), 4
    yield "c4 shape, 3 threads (uneven blocks)", (40.0 * rng.random(4001)).astype(np.float32), 3
    yield "one thread, 8 192 errors (several passes)", (rng.random(8192) * 7.0).astype(np.float32), 1
    yield "two threads", (rng.random(5000) * 1e-3).astype(np.float32), 2
    f = (0.2 + rng.random(4000)).astype(np.float32)
    yield "short mantissas: ties", (f.view(np.uint32) & np.uint32(0xFFFFF800)).view(np.float32), 4
    yield "14-bit mantissas: ties", (f.view(np.uint32) & np.uint32(0xFFFFFC00)).view(np.float32) * np.float32(1024.0), 4
    yield "wide range", np.exp((rng.random(4000) - 0.5) * 40.0).astype(np.float32), 4
    yield "extreme range", np.exp((rng.random(3000) - 0.5) * 150.0).astype(np.float32), 2
    z = (rng.random(4000) * 3.0).astype(np.float32); z[rng.random(4000) < 0.3] = 0.0; z[:300] = 0.0
    yield "zeros", z, 4
    yield "all zero", np.zeros(2048, np.float32), 2
    yield "subnormals", (rng.random(2000) * 1e-39).astype(np.float32), 2
    yield "constant", np.full(4000, 0.7, np.float32), 4
    yield "short blocks", (rng.random(700) * 2.0).astype(np.float32), 4
    yield "one element", np.array([3.25], np.float32), 1
    for name, bad in (("negative", -1.0), ("nan", np.nan), ("inf", np.inf)):
        b = (rng.random(4000) * 5.0).astype(np.float32); b[1234] = bad
        yield name, b, 4


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_wave_chain_equals_the_serial_loop(ctx, lanes):
    for name, e, T in datasets():
        if T > 7 * (64 // lanes):
            continue
        want = serial_sums(e, T)
        wave, lane = ctx.debug_float_chain(e, T, lanes)
        assert wave.view(np.uint32).tolist() == want.view(np.uint32).tolist() or (np.isnan(want).any() and np.array_equal(np.isnan(wave), np.isnan(want))), (name, lanes, wave, want)
        assert lane.view(np.uint32).tolist() == want.view(np.uint32).tolist() or (np.isnan(want).any() and np.array_equal(np.isnan(lane), np.isnan(want))), (name, lanes, lane, want)


def test_many_random_chains(ctx):
    rng = np.random.default_rng(7)
    for trial in range(60):
        T = int(rng.integers(1, 9))
        n = int(rng.integers(T * 260, 8193))
        kind = trial % 4
        if kind == 0: e = rng.random(n) * 10.0 ** rng.integers(-6, 6)
        elif kind == 1: e = np.exp((rng.random(n) - 0.5) * 30.0)
        elif kind == 2:
            e = (0.3 + rng.random(n)).astype(np.float32)
            e = (e.view(np.uint32) & np.uint32(0xFFFFFFFF << int(rng.integers(8, 13)) & 0xFFFFFFFF)).view(np.float32)
        else:
            e = rng.random(n); e[rng.random(n) < 0.5] = 0.0
        e = np.asarray(e, np.float32)
        want = serial_sums(e, T)
        wave, lane = ctx.debug_float_chain(e, T, 32)
        assert wave.view(np.uint32).tolist() == want.view(np.uint32).tolist(), (trial, T, n, kind)
        assert lane.view(np.uint32).tolist() == want.view(np.uint32).tolist(), (trial, T, n, kind)
