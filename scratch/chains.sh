run() { SFQ_MAX_SLOTS=$1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --models $2 --level $3 --block-reads $4 --kernel $5 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('slots',$1,'models',$2,'level',$3,'br',$4,'k',$5,d['ms_per_step'],d['phase_ms']['qlt'],d['phase_ms']['gen'],d['phase_ms']['rec'],d['ratio'])"; }
run 12288 4 1 1024 0
run 65536 4 1 512 0
run 65536 4 1 256 0
run 65536 4 1 256 4
run 65536 4 1 128 0
run 65536 1 3 256 0
run 65536 2 3 256 0
