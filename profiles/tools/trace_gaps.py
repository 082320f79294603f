"""Idle time between consecutive kernels of a rocprofv3 kernel trace (MEASUREMENT TOOL): for the launches of the timed region (the last
`steps` executions of the step, found by the most frequent kernel's count), the gap end(i) -> start(i + 1) on the device timeline.
usage: python profiles/tools/trace_gaps.py t_kernel_trace.csv [launches_per_step]"""
import csv
import sys
import collections


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    # the steady-state region: the last 60 % of the launches
    ev = ev[int(len(ev) * 0.4):]
    busy = sum(e - s for s, e, _ in ev)
    span = ev[-1][1] - ev[0][0]
    gaps = [max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1)]
    overl = sum(1 for i in range(len(ev) - 1) if ev[i + 1][0] < ev[i][1])
    print(f"{len(ev)} launches, span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle between kernels {sum(gaps) / 1e6:.3f} ms "
          f"({100.0 * sum(gaps) / span:.1f} % of the span), overlapping pairs {overl}")
    gs = sorted(gaps)
    print(f"gap per boundary: median {gs[len(gs) // 2] / 1e3:.2f} us, p90 {gs[int(len(gs) * 0.9)] / 1e3:.2f} us, max {gs[-1] / 1e3:.1f} us")
    big = collections.Counter()
    for i, g in enumerate(gaps):
        if g > 20000:
            big[(ev[i][2][:50], ev[i + 1][2][:50])] += 1
    for k, v in big.most_common(8):
        print(f"  > 20 us x{v}: {k[0]} -> {k[1]}")


if __name__ == "__main__":
    main()
