for l in 64 32 16 8; do echo lanes $l; SFQ_DECODE_LANES=$l python scratch/decode_bench.py 10000000 1024 2>&1 | grep "^decode" | tail -1; done
