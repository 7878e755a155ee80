cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/d01
for ex in none allreduce allgather; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-dist --no-overlap --exchange $ex --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$ex', '%.3f ms'%j['ms_per_step'])"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('plain N=1', '%.3f ms'%j['ms_per_step'])"
