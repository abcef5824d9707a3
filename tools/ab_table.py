"""per-build rows of a tools/lib_ab.sh log: python tools/ab_table.py gpurun_out/<tag>/ab.txt"""
import collections
import re
import sys

rows, cur = collections.defaultdict(list), None
for l in open(sys.argv[1]):
    if l.startswith("== "):
        cur = l.split()[1].split("/")[-1]
    m = re.match(r"\S+ +k_lidar_residual +([\d.]+) us \(device clock +([\d.]+)\).*k_lidar_solve +([\d.]+).*visual update +([\d.]+).*8 frames +([\d.]+) ms.*iteration +([\d.]+)", l)
    if m:
        rows[cur].append([float(x) for x in m.groups()])
for k, v in rows.items():
    print("%-30s residual by events %s | device clock %s | solve %s | visual %s | 8 frames ms %s | LiDAR-only wall per iteration %s" % (
        k, *[" ".join("%.2f" % r[i] for r in v) for i in (0, 1, 2, 3)], " ".join("%.3f" % r[4] for r in v), " ".join("%.2f" % r[5] for r in v)))
