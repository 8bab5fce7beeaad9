run() { SFQ_PRIO=$1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel $2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('prio','$1','k',$2,d['value'],d['ms_per_step'],d['phase_ms'])"; }
run 0,0,0,0 0
run -1,-1,1,0 0
run -1,0,1,0 0
run -1,1,0,0 0
run 0,0,-1,0 0
run 0,0,0,0 3
run -1,0,1,0 3
run 1,0,-1,0 3
