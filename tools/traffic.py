"""HBM traffic records of the rocprofv3 --pmc passes (profiles/r*_traffic_<workload>.json) and the rule that ties them to a build.

A record is only quoted by bench.py when it was captured from the SAME device code: every record carries `csrc_sha` (sha256 over fast-livo2_amd/csrc/* and
include/livo2_hip.h, the inputs of liblivo2_hip.so) and `lib_sha` (sha256 of the .so the capture ran); `load()` refuses a record whose `csrc_sha` differs from
the tree's.  (The source hash is the gate because the driver may rebuild the library from the same sources; the .so hash is kept beside it as evidence.)

`python tools/traffic.py make <tag> <workload> <kernel> <points> <fetch.db> <write.db> <tcc.db> [--max]` writes profiles/<tag>_traffic_<workload>.json from the
rocpd databases of three separate --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum), corrected as MI355X_MICROARCH.md prescribes
(FETCH_SIZE is reported in KB and halved on gfx950: x2; cross-check TCC_MISS x 128 B)."""
import glob
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fast-livo2_amd", "lib", "liblivo2_hip.so")


def csrc_sha():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "fast-livo2_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "fast-livo2_amd", "csrc", "*.inc"))) + [os.path.join(ROOT, "include", "livo2_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def lib_sha():
    try:
        with open(LIB, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def load(workload, **match):
    """(HBM bytes per launch or None, note).  Newest profiles/r*_traffic_<workload>.json whose csrc_sha is the tree's and whose keys match."""
    want = csrc_sha()
    stale = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{workload}.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        name = "profiles/" + os.path.basename(path)
        if rec.get("csrc_sha") != want:
            stale.append(name)
            continue
        if all(rec.get(k) == v for k, v in match.items()):
            return (rec["fetch_size_kb_reported"] * rec["fetch_correction"] + rec["write_size_kb"]) * 1024.0, f"{name}: {rec['source']}"
    return None, ("no PMC capture of this build (csrc_sha %s)%s" % (want, "; captures of other builds not quoted: " + ", ".join(stale) if stale else ""))


def committed_trace_us(kernel_prefix, which="c4"):
    """(average kernel duration in us, launches, source) of `kernel_prefix` in the newest committed `rocprofv3 --kernel-trace --stats` summary
    profiles/r*_rocprofv3_kernel_trace_stats_<which>.txt whose capture (profiles/r*_build_sha.txt of the same tag) ran THIS device code; (None, None, note) otherwise.
    bench.py quotes it BESIDE its own live event timing (another run, possibly another box): the kernel's own duration against the event bracket around its launch."""
    want = csrc_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_rocprofv3_kernel_trace_stats_{which}.txt")), reverse=True):
        tag = os.path.basename(path).split("_")[0]
        try:
            built = open(os.path.join(ROOT, "profiles", f"{tag}_build_sha.txt")).read().split()[0]
        except (OSError, IndexError):
            continue
        if built != want:
            continue
        for ln in open(path):
            if ln.startswith("#"):
                continue
            m = re.match(r"^(.*), (\d+), ([0-9.]+), ([0-9.]+), ([0-9.]+) \[", ln)          # name (may contain commas), calls, avg_us, min_us, max_us [...]
            if m and kernel_prefix in m.group(1):
                return float(m.group(3)), int(m.group(2)), "profiles/" + os.path.basename(path)
    return None, None, "no committed kernel trace of this build (csrc_sha %s)" % want


def _counter(db, kernel, counter, use_max):
    c = sqlite3.connect(db)
    agg = "max" if use_max else "avg"
    pattern = "%" + kernel + ("<%" if kernel == "k_lidar_residual" else "%")          # (k_lidar_residual<256> must not pick up k_lidar_residual_batch)
    row = c.execute(f"select {agg}(value), count(*) from counters_collection where kernel_name like ? and counter_name = ?", (pattern, counter)).fetchone()
    if row is None or row[0] is None:
        raise SystemExit(f"traffic.py: no {counter} for {kernel} in {db}")
    return float(row[0]), int(row[1])


def make(tag, workload, kernel, points, fetch_db, write_db, tcc_db, use_max=False, command=""):
    fetch, n = _counter(fetch_db, kernel, "FETCH_SIZE", use_max)
    write, _ = _counter(write_db, kernel, "WRITE_SIZE", use_max)
    miss, _ = _counter(tcc_db, kernel, "TCC_MISS_sum", use_max)
    rec = {"kernel": kernel, "points": int(points), "fetch_size_kb_reported": fetch, "fetch_correction": 2.0, "write_size_kb": write,
           "tcc_miss_x128B_MB": miss * 128.0 / 1e6, "dispatches": n, "aggregate": "max per dispatch" if use_max else "mean per dispatch",
           "csrc_sha": csrc_sha(), "lib_sha": lib_sha(),
           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_* passes (separate runs, --kernel-trace only) of `{command}`; FETCH_SIZE x2 per MI355X_MICROARCH.md, "
                     f"cross-check TCC_MISS x 128 B = {miss * 128.0 / 1e6:.1f} MB"}
    out = os.path.join(ROOT, "gpurun_out", f"{tag}_traffic_{workload}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    print(out, json.dumps(rec))


WORKLOADS = {            # workload -> (kernel, where bench_full.json holds the points of one launch, aggregate)
    "c4": ("k_lidar_residual", ("config", "points_per_frame"), False),
    "c4_lockstep": ("k_lidar_residual_batch", ("extra", "c4_lockstep", "roofline", "bytes_per_launch"), False),
    "batched": ("k_lidar_residual_batch", ("extra", "batched", "points_per_launch"), False),
    "out_of_cache": ("k_lidar_residual_batch", ("extra", "out_of_cache", "points_per_launch"), True),      # the 64-frame launches are the largest dispatches of the run
}


def make_from_bench(tag, workload, fetch_db, write_db, tcc_db, bench_full, command):
    kernel, path, use_max = WORKLOADS[workload]
    v = json.load(open(bench_full))
    for k in path:
        v = v[k]
    points = int(round(v / 276.0)) if path[-1] == "bytes_per_launch" else int(v)
    make(tag, workload, kernel, points, fetch_db, write_db, tcc_db, use_max, command)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "make-from-bench":
        make_from_bench(a[1], a[2], a[3], a[4], a[5], a[6], a[7] if len(a) > 7 else "")
    elif a and a[0] == "sha":
        print(csrc_sha(), lib_sha())
    elif a and a[0] == "make":
        use_max = "--max" in a
        a = [x for x in a if x != "--max"]
        make(a[1], a[2], a[3], int(a[4]), a[5], a[6], a[7], use_max, a[8] if len(a) > 8 else "")
    else:
        raise SystemExit(__doc__)
