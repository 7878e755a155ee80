"""Summarises the weighted-median kernels of a rocprofv3 kernel trace of `bench.py --pp` (debug aid; scripts/gpu_wm1.sh): the
last psm_wgt_median call's launches in order - start offset, duration, gap to the previous one - and the totals per kernel."""
import glob, sqlite3, sys

dbs = glob.glob(sys.argv[1] + "/prof/**/*.db", recursive=True)
if not dbs:
    print("no rocprofv3 database under", sys.argv[1]); sys.exit(0)
rows = list(sqlite3.connect(dbs[0]).execute("select name, start, end from kernels order by start"))
seeds = [i for i, r in enumerate(rows) if "k_wm_seed" in r[0]]
if not seeds:
    print("no weighted-median kernels in the trace"); sys.exit(0)
first = seeds[-1]
while first - 1 in seeds:          # (round 5's form: one seed launch per map)
    first -= 1
run = rows[first:]
t0 = run[0][1]
print(f"{len(run)} launches, span {(run[-1][2] - t0) / 1e6:.3f} ms, busy {sum(e - s for _, s, e in run) / 1e6:.3f} ms")
agg, prev = {}, None
for n, s, e in run:
    short = n.split("(")[0].replace("void psm::", "").replace("psm::", "")
    a = agg.setdefault(short, [0, 0])
    a[0] += 1; a[1] += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {((s - prev) / 1e3 if prev else 0):6.1f}  {short}")
    prev = e
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:50s} {n:4d} launches {t / 1e6:8.3f} ms")
