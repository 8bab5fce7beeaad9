run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline --models $1 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('models',$1,d['ms_per_step'],d['phase_ms'])"; }
run 3; run 5; run 6; run 1; run 2; run 4; run 7
