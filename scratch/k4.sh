for k in 2 4 8; do SFQ_GEN_CHAINS=$k python bench.py --steps 3 --warmup 1 --no-cpu-baseline --models 2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('gen chains',$k,'alone',d['ms_per_step'])"
SFQ_GEN_CHAINS=$k python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('gen chains',$k,'all',d['value'],d['ms_per_step'],d['phase_ms'])"
done
