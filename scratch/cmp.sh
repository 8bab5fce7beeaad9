run() { SFQ_GRID_Q=$1 SFQ_GRID_G=$2 SFQ_GRID_R=$3 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('grid',$1,$2,$3,d['value'],d['ms_per_step'],d['phase_ms']['qlt'],d['phase_ms']['gen'],d['phase_ms']['rec'])"; }
run 99999 99999 99999
run 5456 5456 99999
run 99999 99999 99999
run 5456 5456 99999
