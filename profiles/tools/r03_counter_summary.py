#!/usr/bin/env python3
"""Per-kernel means of the --pmc passes of r03_front_counters.sh (one JSON on stdout)."""
import collections, csv, glob, json, re, sys
O = sys.argv[1]
out = collections.defaultdict(dict)
for d in ("sq1", "sq2", "tcp", "tcc"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (O, d), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+)", r["Kernel_Name"])
            if not m:
                continue
            name = m.group(1)
            # variants of one template differ in their template arguments: keep them apart
            t = re.search(r"(k_\w+<[^>]*>)", r["Kernel_Name"])
            agg[t.group(1) if t else name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "VGPR_Count" in r:
                out[t.group(1) if t else name]["vgpr"] = r.get("VGPR_Count") or r.get("Arch_VGPR_Count")
        for k, cs in agg.items():
            for cn, v in cs.items():
                out[k][cn] = sum(v) / len(v)
                out[k]["launches"] = len(v)
for f in glob.glob("%s/kt/**/*kernel_stats.csv" % O, recursive=True):
    for r in csv.DictReader(open(f)):
        t = re.search(r"(k_\w+<[^>]*>)", r["Name"]) or re.search(r"(k_\w+)", r["Name"])
        if t:
            out[t.group(1)]["avg_ns"] = float(r["AverageNs"])
json.dump(out, sys.stdout, indent=1, sort_keys=True)
