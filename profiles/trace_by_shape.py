"""Group a rocprofv3 kernel_trace.csv by (kernel, grid) and print per-launch averages.
usage: python profiles/trace_by_shape.py <kernel_trace.csv> [steps]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    short = (m.group(1) if m else name)[:58]
    wg = int(r["Workgroup_Size_X"]); grid = int(r["Grid_Size_X"]) // max(wg, 1)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault((short, grid), [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
nl = sum(v[0] for v in agg.values())
lib = sum(v[0] for k, v in agg.items() if not (k[0].startswith("at::") or k[0].startswith("__amd_rocclr") or k[0].startswith("rocprim") or "at::native" in k[0]))
print(f"total kernel time {tot/1e3:.3f} ms over {steps} steps = {tot/1e3/steps:.3f} ms/step")
print(f"launches per step {nl/steps:.1f} (this library's kernels {lib/steps:.1f}, torch / runtime {(nl-lib)/steps:.1f}); every (kernel, grid) pair is listed")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:60s} wgs={k[1]:7d} n/step={v[0]/steps:5.1f} avg_us={v[1]/v[0]:8.1f} ms/step={v[1]/1e3/steps:7.3f} share={100*v[1]/tot:5.1f}%")
