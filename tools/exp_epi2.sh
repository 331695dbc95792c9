cd $GRAFT_REPO_ROOT
for d in 0 1 0 1; do echo "== direct $d"; OVHIP_GEMM_EPI_DIRECT=$d python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --breakdown 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.readline()); print(o['value'], o['ms_per_step'], {k:(v['ms'] if isinstance(v,dict) else v) for k,v in o['breakdown'].items()})"; done
