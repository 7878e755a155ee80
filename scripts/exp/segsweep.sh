cd $GRAFT_REPO_ROOT
for sr in 90 108 135 180 216 270 360 540 1080; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shard-sim 8 --seg-rows $sr 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('shard 1/8 seg $sr', '%.3f ms'%j['ms_per_step'], j['kernels']['cvf_fused']['avg_ms'])"
done
