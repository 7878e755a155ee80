# world_size-1 runs of the torch.distributed/RCCL code path + shard simulations (one GPU).  bash scripts/exp/dist_exp.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/d01
run() { # label, args
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline $2 2>gpurun_out/d01/err_$1.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', '%.3f ms'%j['ms_per_step'])"
}
run pipelined ""
run overlap "--no-frame-pipeline"
run nooverlap "--no-overlap"
run allgather "--exchange allgather"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('plain N=1', '%.3f ms'%j['ms_per_step'])"
for g in 2 4 8; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shard-sim $g 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('shard-sim 1/$g', '%.3f ms'%j['ms_per_step'], {k:v['avg_ms'] for k,v in j['kernels'].items()})"
done
