"""Per-kernel wave-cycle breakdown from one rocprofv3 PMC pass (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU): parked at s_waitcnt / s_barrier, issue-stalled, issuing.
usage: python profiles/pmc_sq_breakdown.py <counter_collection.csv>"""
import collections, csv, re, sys
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    key = (m.group(1) if m else name)[:64]
    a = agg.setdefault(key, collections.defaultdict(float))
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        a["n"] += 1
print(f"{'kernel':64s} {'n':>4s} {'parked%':>8s} {'stall%':>7s} {'issue%':>7s} {'valu%':>6s} {'lds%':>5s} {'ldsconf/wavecyc':>15s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:16]:
    wc = max(a["SQ_WAVE_CYCLES"], 1.0)
    print(f"{k:64s} {int(a['n']):4d} {100*a['SQ_WAIT_ANY']/wc:8.1f} {100*a['SQ_WAIT_INST_ANY']/wc:7.1f} {100*a['SQ_ACTIVE_INST_ANY']/wc:7.1f} "
          f"{100*a['SQ_ACTIVE_INST_VALU']/wc:6.1f} {100*a['SQ_ACTIVE_INST_LDS']/wc:5.1f} {a['SQ_LDS_BANK_CONFLICT']/wc:15.4f}")
