"""Per-kernel time and inter-kernel gaps from a rocprofv3 kernel-trace csv (steady-state part).
    python scripts/trace_gaps.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in rows))[skip:]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    gaps[(n0[-28:], n1[-28:])].append(s1 - e0)
print(f"kernels {len(ev)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {100*(span-busy)/span:.1f} %")
per = collections.defaultdict(list)
for s, e, n in ev:
    per[n].append(e - s)
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:62s} n={len(v):4d} avg {sum(v)/len(v)/1e3:9.1f} us  total {sum(v)/1e6:8.3f} ms")
print("gaps (us, avg) between consecutive kernels:")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  {k[0]:>28s} -> {k[1]:28s} n={len(v):4d} avg {sum(v)/len(v)/1e3:8.1f}")
