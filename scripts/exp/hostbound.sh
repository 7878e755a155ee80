cd $GRAFT_REPO_ROOT
for cfg in c2 c3; do
for extra in "" "--no-frame-pipeline" "--no-overlap" "--exchange none --no-overlap"; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-dist --config $cfg --steps 50 --warmup 5 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$cfg dist [$extra]', '%.3f ms'%j['ms_per_step'], {k:v['avg_ms'] for k,v in j['kernels'].items()})"
done
timeout 300 python bench.py --config $cfg --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$cfg plain', '%.3f ms'%j['ms_per_step'])"
done
