#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals for the trailing ADMM part of a bench
run (after the last L-BFGS kernel) or, with `alm` as second argument, for the phase-1 part before it;
busy fraction and gaps.  usage: trace_summary.py trace.csv [alm]"""
import collections
import csv
import re
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


# kernels only phase 1 launches: the ADMM part of a bench run starts behind the last of them (k_his_two*: the launch-by-launch inner
# iteration; k_lbfgs_team / k_alm_close: round 4's fused one)
PHASE1_MARKS = ("k_his_two", "k_lbfgs_team", "k_alm_close")


def short(n):
    m = re.search(r"(k_\w+|__amd_\w+)", n)
    return m.group(1) if m else n[:40]


last = max(i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in PHASE1_MARKS))
alm = len(sys.argv) > 2 and sys.argv[2] == "alm"
adm = rows[:last + 1] if alm else rows[last + 1:]
if alm:
    print("inner iterations (k_his_two / k_lbfgs_team launches): %d" % sum(1 for r in adm if "k_his_two" in r["Kernel_Name"] or "k_lbfgs_team" in r["Kernel_Name"]))
t0 = int(adm[0]["Start_Timestamp"])
t1 = int(adm[-1]["End_Timestamp"])
print(("ALM-part" if alm else "ADMM-part") + " kernels: %d  span %.3f ms" % (len(adm), (t1 - t0) / 1e6))
agg = collections.OrderedDict()
for r in adm:
    k = short(r["Kernel_Name"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0, []])
    a[0] += 1
    a[1] += d
    a[2].append(d)
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%-30s calls %5d  total %9.1f us  avg %7.2f us  med %7.2f us  %5.1f%%" %
          (k, v[0], v[1] / 1e3, v[1] / v[0] / 1e3, statistics.median(v[2]) / 1e3, 100 * v[1] / tot))
print("sum of kernel time %.3f ms, busy fraction %.2f" % (tot / 1e6, tot / (t1 - t0)))
gaps = [int(adm[i + 1]["Start_Timestamp"]) - int(adm[i]["End_Timestamp"]) for i in range(len(adm) - 1)]
print("gaps: median %.2f us  mean %.2f us  max %.1f us  (>20us: %d)" %
      (statistics.median(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3, max(gaps) / 1e3, sum(1 for g in gaps if g > 20000)))
