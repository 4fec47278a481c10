cd $GRAFT_REPO_ROOT
for P in 1 2 3 4 7; do
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$P bench.py --gpus 1 --sharded --no-c4-leg --steps 5 --warmup 2 --pieces $P 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pieces', $P, round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'one_gpu', round(d['one_gpu_same_run_value']/1e6,2), d['gather_pieces'])"
done
