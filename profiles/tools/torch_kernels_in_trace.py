"""List the non-library kernels (torch / runtime) of a rocprofv3 kernel trace with their full names, launches per step and grid.
usage: python profiles/tools/torch_kernels_in_trace.py <kernel_trace.csv> <steps>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "(anonymous namespace)" in n and "at::" not in n:
        continue
    key = (n[:230], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[0] for v in agg.values())
print(f"{tot / steps:.1f} torch/runtime launches per step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"n/step={v[0]/steps:6.1f} avg_us={v[1]/v[0]:6.1f} wgs={k[1]:6d}  {k[0]}")
