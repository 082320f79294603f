"""Group a rocprofv3 kernel_trace.csv by (kernel, grid) and print per-launch averages.
usage: python profiles/trace_by_shape.py <kernel_trace.csv> [steps | auto]
steps = auto: the number of EXECUTIONS of the step is counted from the trace itself - the greatest common divisor of the launch counts of
this library's (kernel, grid) pairs (every execution of a forward launches each of them the same number of times; warm-up and eager
legs are executions too) - so that n/step comes out as an integer.  (r03's files divided by the --steps argument and printed 5.4 / 2.2.)"""
import collections, csv, math, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = sys.argv[2] if len(sys.argv) > 2 else "1"
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    short = (m.group(1) if m else name)[:58]
    # workgroups of the launch = the product over all three grid dimensions (r03's files printed Grid_Size_X only: the split-K weight
    # gradients, whose splits are in grid.y, read as "4-16 workgroups" when the launches have 256-512)
    dims = [int(r.get("Grid_Size_" + d) or 1) // max(int(r.get("Workgroup_Size_" + d) or 1), 1) for d in "XYZ"]
    grid = "x".join(str(d) for d in dims if d > 1) or "1"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault((short, grid), [0, 0.0])
    a[0] += 1; a[1] += d
is_lib = lambda name: not (name.startswith("at::") or name.startswith("__amd_rocclr") or name.startswith("rocprim") or "at::native" in name
                           or name.startswith("split_weights") or name.startswith("convert"))
if steps == "auto":
    steps = 0
    for k, v in agg.items():
        if is_lib(k[0]) and v[0] >= 2:
            steps = math.gcd(steps, v[0])
    steps = max(steps, 1)
else:
    steps = int(steps)
tot = sum(v[1] for v in agg.values())
nl = sum(v[0] for v in agg.values())
lib = sum(v[0] for k, v in agg.items() if not (k[0].startswith("at::") or k[0].startswith("__amd_rocclr") or k[0].startswith("rocprim") or "at::native" in k[0]))
print(f"total kernel time {tot/1e3:.3f} ms over {steps} executions of the step = {tot/1e3/steps:.3f} ms/step")
print(f"launches per step {nl/steps:.1f} (this library's kernels {lib/steps:.1f}, torch / runtime {(nl-lib)/steps:.1f}); every (kernel, grid) pair is listed")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:60s} wgs={k[1]:>9s} n/step={v[0]/steps:5.1f} avg_us={v[1]/v[0]:8.1f} ms/step={v[1]/1e3/steps:7.3f} share={100*v[1]/tot:5.1f}%")
