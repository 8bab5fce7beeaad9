run() { SFQ_GRID_Q=$1 SFQ_GRID_G=$2 SFQ_GRID_R=$3 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel $4 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('grid',$1,$2,$3,'k',$4,d['value'],d['ms_per_step'],d['phase_ms'])"; }
run 3072 2560 2560 0
run 4096 2048 2048 0
run 2560 2560 3072 0
run 2048 2048 2048 0
run 3072 2560 2560 3
run 4096 4096 4096 3
run 2560 2560 3072 3
