"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes).  gfx950 corrections: counters are in KiB; FETCH_SIZE under-reports wide
coalesced reads by exactly 2x, so read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is exact.
usage: python profiles/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]"""
import collections, csv, json, re, sys


def per_kernel(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
        key = (m.group(1) if m else name)[:70]
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    n, f = fetch[k]
    w = write.get(k, [n, 0.0])[1]
    out[k] = {"launches": n, "read_bytes_per_launch": f * 1024 * 2 / n, "write_bytes_per_launch": w * 1024 / write.get(k, [n])[0],
              "hbm_bytes_per_launch": f * 1024 * 2 / n + w * 1024 / write.get(k, [n])[0]}
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:25]:
    print(f"{k:70s} n={v['launches']:4d} read={v['read_bytes_per_launch']/1e6:9.2f} MB write={v['write_bytes_per_launch']/1e6:9.2f} MB")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
