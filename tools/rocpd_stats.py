"""Per-kernel statistics from a rocprofv3 `--kernel-trace` run stored as a rocpd
SQLite database (the default output of ROCm 7.x), in the column layout of the
profiler's own `kernel_stats.csv`.

    python tools/rocpd_stats.py results.db [out.csv]
"""
import csv
import math
import sqlite3
import sys


def kernel_stats(db_path):
    con = sqlite3.connect(db_path)
    rows = con.execute("select name, duration from kernels").fetchall()
    by = {}
    for name, d in rows:
        by.setdefault(name, []).append(int(d))
    total = sum(sum(v) for v in by.values()) or 1
    out = []
    for name, v in by.items():
        n, s = len(v), sum(v)
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
        out.append((name, n, s, round(mean, 6), round(100.0 * s / total, 4), min(v), max(v), round(sd, 6)))
    out.sort(key=lambda r: -r[2])
    return out


def main():
    stats = kernel_stats(sys.argv[1])
    f = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    w.writerows(stats)


if __name__ == "__main__":
    main()
