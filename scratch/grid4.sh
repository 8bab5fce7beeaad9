run() { SFQ_GRID_Q=$1 SFQ_GRID_G=$2 SFQ_GRID_R=$3 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('grid',$1,$2,$3,d['value'],d['ms_per_step'],d['phase_ms']['qlt'],d['phase_ms']['gen'],d['phase_ms']['rec'])"; }
run 5456 5456 2728
run 6000 6000 2000
run 6400 6400 1600
run 5600 6400 2000
run 6400 5600 2000
run 4800 4800 3200
run 7000 7000 1200
