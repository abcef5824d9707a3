"""The driver parses ONE JSON line from bench.py's stdout.  BENCH_r03.json.parsed was null because that line had grown to 22 KB: the line is now a compact
headline (< 4 KB, strict JSON: no NaN / Infinity tokens) and everything else goes to gpurun_out/bench_full.json + stderr.  `--emit-selftest` runs exactly the
emission code of the real run on a report of the real shape."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_constants(tok):
    raise ValueError("non-strict JSON token " + tok)


def test_headline_is_one_small_strict_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emit-selftest"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                        # nothing but the headline on stdout
    assert len(lines[0].encode()) < 4096
    out = json.loads(lines[0], parse_constant=_no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "pre_warm_s"):
        assert k in out, k
    assert "workload" in out["config"] and "model" not in out["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_us"):
        assert k in out["roofline"], k
    # the live device-clock figures (round 6) come before anything read back from a committed capture, and the box hint is in the line
    keys = list(out["roofline"])
    for k in ("kernel_us_device", "frac_device", "per_rank_lidar_solve_us", "box_kind"):
        assert k in keys, k
    assert keys.index("kernel_us_device") < keys.index("kernel_us_rocprofv3") and keys.index("frac_device") < keys.index("frac_rocprofv3")
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    assert "extra" not in out
    # the full report: strict JSON too, on stderr and in gpurun_out/
    full_lines = [ln for ln in r.stderr.splitlines() if ln.startswith("bench.py full report: ")]
    assert len(full_lines) == 1
    full = json.loads(full_lines[0][len("bench.py full report: "):], parse_constant=_no_constants)
    assert full["extra"]["nan"] is None and full["extra"]["inf"] is None and full["extra"]["arr"] == [0, 1, 2]
    assert len(full["extra"]["big_note"]) == 65536


def test_traffic_records_are_tied_to_the_build():
    """a PMC capture is only quoted for the device code it was taken from (tools/traffic.py)"""
    sys.path.insert(0, ROOT)
    from tools import traffic
    import glob
    want = traffic.csrc_sha()
    for wl in ("c4", "c4_lockstep", "batched", "out_of_cache"):
        val, note = traffic.load(wl)
        matching = [p for p in glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{wl}.json")) if json.load(open(p)).get("csrc_sha") == want]
        assert (val is not None) == bool(matching), (wl, note)
        if val is None:
            assert "no PMC capture of this build" in note


def test_committed_kernel_trace_is_quoted_only_for_the_build_it_was_taken_from():
    """bench.py puts the rocprofv3 kernel-only duration of k_lidar_residual from the committed trace BESIDE its live event timing (roofline.kernel_us_rocprofv3 /
    frac_rocprofv3) — under the same rule as the traffic records: only a capture of THIS device code (tools/traffic.py::committed_trace_us)"""
    sys.path.insert(0, ROOT)
    from tools import traffic
    import glob
    want = traffic.csrc_sha()
    us, launches, src = traffic.committed_trace_us("k_lidar_residual<")
    tags = [os.path.basename(p).split("_")[0] for p in glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_trace_stats_c4.txt"))]
    matching = [t for t in tags if os.path.exists(os.path.join(ROOT, "profiles", t + "_build_sha.txt")) and open(os.path.join(ROOT, "profiles", t + "_build_sha.txt")).read().split()[0] == want]
    assert (us is not None) == bool(matching), src
    if us is not None:
        assert 5.0 < us < 100.0 and launches > 100 and src.startswith("profiles/")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emit-selftest"], capture_output=True, text=True, timeout=300)
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
        assert out["roofline"]["kernel_us_rocprofv3"] == us and out["roofline"]["rocprofv3_src"] == src
        assert abs(out["roofline"]["frac_rocprofv3"] - 276.0 * 200000 / (us * 1e-6) / 8e12) < 1e-9
    else:
        assert "no committed kernel trace of this build" in src
