run() { SFQ_ORDER=$1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel $2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('order','$1','k',$2,d['value'],d['ms_per_step'],d['phase_ms'])"; }
for o in qgr rgq rqg grq gqr qrg; do run $o 0; done
run rgq 4
