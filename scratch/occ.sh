run() { SFQ_GRID_Q=$1 SFQ_GRID_G=$1 SFQ_GRID_R=$1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel 3 --models $2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('grid',$1,'models',$2,d['ms_per_step'],d['phase_ms'])"; }
for m in 4 2 1; do for g in 8192 4096 2048 1024; do run $g $m; done; done
