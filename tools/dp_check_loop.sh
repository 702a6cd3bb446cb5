#!/bin/bash
# the 2-rank same-device dry run of bench.py N times: dp_check of each run (hunting a timing-dependent mismatch)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in $(seq 1 ${1:-5}); do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700 + i)) bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --same-device 2>/dev/null \
    | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)['train_step']['dp_check']
        print({k: d[k] for k in ('reduced_equals_sum_of_local_rel_err', 'worst_element', 'local_backward_repeat', 'params_identical_on_all_ranks', 'task')})
"
done
