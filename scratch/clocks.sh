#!/bin/bash
# engine clock / power while the default bench runs (is the co-run phase power- or clock-limited?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|mclk" | head -6
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-adaptive-leg --no-decode > /tmp/b.log 2>&1 &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Average Graphics Package Power\|Current Socket\|power (W)" | tr '\n' ' '; echo; sleep 1; done
wait $BP
tail -1 /tmp/b.log | cut -c1-160
