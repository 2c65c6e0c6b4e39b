cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in 1 0; do
ms=$(CMR_AGENT_UPDATE_FORK=$v timeout -k 10 200 python bench.py --mode train --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'])")
echo "rep $rep towers forked=$v (ROC_CPU_WAIT_FOR_SIGNAL=1) -> $ms" | tee -a gpurun_out/r06_ab_update_fork.txt
done; done
