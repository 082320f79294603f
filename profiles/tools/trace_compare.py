"""Compare two rocprofv3 kernel traces of the same command: launches and GPU time per kernel family (name up to the template list,
plus the grid size), per step.  usage: python profiles/tools/trace_compare.py A_kernel_trace.csv B_kernel_trace.csv [steps=8] [by=name|shape]"""
import collections
import csv
import re
import sys


def load(path, by):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        n = re.sub(r"<.*", "", n) if by == "name" else n[:70]
        if by == "shape":
            n += f" wgs={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}"
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[n][0] += 1
        agg[n][1] += d
    return agg


def main():
    a_path, b_path = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    by = sys.argv[4] if len(sys.argv) > 4 else "name"
    a, b = load(a_path, by), load(b_path, by)
    ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
    na, nb = sum(v[0] for v in a.values()), sum(v[0] for v in b.values())
    print(f"per step: launches {na / steps:.1f} -> {nb / steps:.1f}; kernel time {ta / steps / 1e3:.3f} -> {tb / steps / 1e3:.3f} ms")
    keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, [0, 0])[1] - b.get(k, [0, 0])[1]))
    for k in keys[:40]:
        x, y = a.get(k, [0, 0.0]), b.get(k, [0, 0.0])
        print(f"{x[0] / steps:7.1f} -> {y[0] / steps:7.1f} launches  {x[1] / steps / 1e3:8.3f} -> {y[1] / steps / 1e3:8.3f} ms  ({(y[1] - x[1]) / steps / 1e3:+.3f})  {k}")


if __name__ == "__main__":
    main()
