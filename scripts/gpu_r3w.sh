#!/bin/bash
# Round-3 session W: weighted median - parity, timing, kernel traces of the two representative inputs
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3w
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=1200 -k "wmf or wgt or median or pp or process_dm or fuzz" 2>&1 | tail -4
timeout 300 python scripts/dbg_wmf.py big 2>&1 | tee $OUT/wmf.txt | tail -12
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --frame-loop 0 --pp > $OUT/c4pp.json 2>$OUT/err; python -c "
import json;j=json.loads([l for l in open('$OUT/c4pp.json') if l.startswith('{')][-1]);print(j['pp'])"
timeout 200 python scripts/soak.py 60 ${SOAK_SEED:-54} | tail -2
cd /tmp; export TMPDIR=/tmp
for which in hd20 pp; do
if [ $which = hd20 ]; then CMD="python $GRAFT_REPO_ROOT/scripts/dbg_wmf.py hd20"; else CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --frame-loop 0 --pp"; fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$which -o trace -- $CMD > $OUT/trace_$which.log 2>&1
f=$(find $OUT/prof_$which -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_wm_seed" in r["Kernel_Name"]]
start=idx[-2]
end=max(i for i,r in enumerate(rows) if "k_wm_" in r["Kernel_Name"])
t0=int(rows[start]["Start_Timestamp"]); prev_end=t0; out=[]
for r in rows[start:end+1]:
    n=r["Kernel_Name"]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    short=n.split("(")[0].replace("void psm::","").replace("psm::","")[:40]
    out.append((short,(s-t0)/1e3,(e-s)/1e3,(s-prev_end)/1e3)); prev_end=e
tot={}; cnt={}
for sh,st,du,gap in out: tot[sh]=tot.get(sh,0)+du; cnt[sh]=cnt.get(sh,0)+1
print("$which: total span us", (prev_end-t0)/1e3, "sum kernels", sum(tot.values()), "gaps", sum(max(g,0) for _,_,_,g in out), "launches", len(out))
for k,v in sorted(tot.items(), key=lambda kv:-kv[1]): print(f"  {k:42s} {v:9.1f} us  x{cnt[k]}")
PY
done
find $OUT -name "*.csv" -size +3M -delete
