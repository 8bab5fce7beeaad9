# the other BASELINE configurations, one JSON line each
python bench.py --steps 3 --warmup 1 --workload qlt --no-cpu-baseline 2>/dev/null | tail -1
python bench.py --steps 3 --warmup 1 --kind 2 --no-cpu-baseline 2>/dev/null | tail -1
python bench.py --steps 3 --warmup 1 --kind 3 --no-cpu-baseline 2>/dev/null | tail -1
python bench.py --steps 3 --warmup 1 --kind 1 --cpu-sample-reads 2000 2>/dev/null | tail -1
python bench.py --steps 3 --warmup 1 --level 4 --no-cpu-baseline 2>/dev/null | tail -1
