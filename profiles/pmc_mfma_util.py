"""Per-kernel matrix-core utilisation from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE).
util = MFMA busy cycles summed over the chip / (kernel cycles x 1024 SIMDs); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
(the counter is summed over the XCDs).  usage: python profiles/pmc_mfma_util.py <counter_collection.csv> [out.json]"""
import collections, csv, json, re, sys

agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    key = (m.group(1) if m else name)[:70]
    a = agg.setdefault(key, collections.defaultdict(float))
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        a["n"] += 1; a["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for k, a in agg.items():
    n = max(a["n"], 1)
    cyc = a["GRBM_GUI_ACTIVE"] / 8 / n
    out[k] = {"launches": int(n), "avg_us": a["ns"] / n / 1e3, "mfma_busy_cycles_per_launch": a["SQ_VALU_MFMA_BUSY_CYCLES"] / n,
              "kernel_cycles": cyc, "mfma_util_pct": 100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (cyc * 1024) if cyc else 0.0}
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])[:20]:
    print(f"{k:62s} n={v['launches']:3d} avg_us={v['avg_us']:8.1f} mfma_util={v['mfma_util_pct']:5.1f}%")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
