run() { SFQ_MAX_SLOTS=$1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --models $2 --block-reads $3 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('maxslots',$1,'models',$2,'br',$3,d['ms_per_step'],d['config']['blocks_per_gpu'],d['ratio'])"; }
run 12288 2 1024
run 32768 2 512
run 65536 2 256
run 12288 1 1024
run 32768 1 512
run 12288 4 1024
run 12288 4 512
