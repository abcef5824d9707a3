"""the numbers DESIGN.md section 6 / README / BASELINE.md quote, from profiles/<tag>_bench_full.json and the kernel trace: python tools/capture_numbers.py [r06]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
b = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_full.json")))
r, cb, e = b["roofline"], b["cpu_baseline"], b["extra"]
print("sha", open(os.path.join(ROOT, "profiles", f"{tag}_build_sha.txt")).read().strip(), "| suite:", open(os.path.join(ROOT, "profiles", f"{tag}_pytest_gpu.txt")).read().strip().splitlines()[-1])
print("value %.4e  ms/step %.4f  ms/frame %.4f" % (b["value"], b["ms_per_step"], b["ms_per_step"] / 8))
print("residual: device %.2f us (%.3f)  events %.2f (%.3f)  traffic %.1f MB  solve events %s" % (r["kernel_us_device"], r["frac_device"], r["kernel_us"], r["frac"], r["traffic"] / 1e6, r["per_rank_lidar_solve_us"]))
for line in open(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_kernel_trace_stats_c4.txt")):
    m = re.match(r"(void )?(k_lidar_residual<256>|k_lidar_solve|k_visual_update_persistent<false>)[^,]*,.*?, (\d+), ([\d.]+),", line)
    if m:
        print("trace", m.group(2), m.group(3), "launches", m.group(4), "us", ("frac %.3f" % (55.2e6 / (float(m.group(4)) * 1e-6) / 8e12)) if "residual" in m.group(2) else "")
print("cpu %.3e (lidar %.0f ms + visual %.0f ms)  1 thread %.2e  all cores %.2e  oracle live %.1f ms" % (cb["value"], cb["lidar_update_ms"], cb["visual_update_ms"], cb["value_1thread"], cb["value_all_cores"], cb["live_chain"]["avia"]["ms_per_frame"]))
for k, v in e["live_chain"].items():
    if k == "def":
        continue
    for m in ("lean", "full"):
        if m in v:
            print("live", k, m, {a: c for a, c in v[m].items() if "ms" in a and "outside" not in a})
for k, v in e["c5"].items():
    if isinstance(v, dict):
        print("c5", k, "one ctx %.0f  three %.0f" % (v.get("frames_per_s_frame_api_one_context", 0), v.get("frames_per_s_frame_api", 0)))
for k, v in e["map_update"].items():
    if isinstance(v, dict):
        print("map_update", k, "call %.3f ms  kernels %.0f us" % (v.get("map_update_ms_median", 0), v.get("map_update_kernel_us_median", 0)))
L = e["c4_lockstep"]
print("lockstep %.3e frac %.3f traffic %.1f MB" % (L["evals_per_s"], L["roofline"]["frac"], L["roofline"]["traffic"] / 1e6))
print("batched %.3e frac %.3f | ooc %.3e frac %.3f %.2f TB/s" % (e["batched"]["evals_per_s"], e["batched"]["roofline"]["frac"], e["out_of_cache"]["evals_per_s"], e["out_of_cache"]["roofline"]["frac"], e["out_of_cache"]["roofline"]["traffic_GBps"] / 1e3))
print("chains 2: %.3e 4: %.3e" % (e["c4_concurrent_chains"]["2_chains"]["evals_per_s"], e["c4_concurrent_chains"]["4_chains"]["evals_per_s"]))
print("V6", {k: round(v["us_per_update"], 1) for k, v in e["visual_inverse"].items() if isinstance(v, dict)})
print("prestage %.1f us  imu %.1f us (cpu %.1f)  lio_frame %.3f ms  frames_per_s_live %.0f" % (e["preprocess_scan"]["kernel_us"], e["imu_propagate"]["kernel_us"], cb["imu_propagate_us_20_samples"], e["lio_frame"]["one_call_ms"], e["frames_per_s_live"]["value"]))
