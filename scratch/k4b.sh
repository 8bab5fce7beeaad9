for k in 2 4; do SFQ_GEN_CHAINS=$k python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('gen chains',$k,d['value'],d['ms_per_step'],d['phase_ms']['qlt'],d['phase_ms']['gen'],d['phase_ms']['rec'])"; done
