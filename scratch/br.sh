for br in 1024 1221 1300 2442; do for k in 0 3; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --block-reads $br --kernel $k 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('br',$br,'k',$k,d['value'],d['ms_per_step'],d['phase_ms'],d['ratio'])"
done; done
