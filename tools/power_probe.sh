#!/bin/bash
# Clock / power while one convolution kernel runs back to back:  tools/power_probe.sh N Cin Cout HW wino|direct
cd $GRAFT_REPO_ROOT
python -u tools/wino_one.py "$@" 3 > /tmp/pp0.log 2>&1     # warm the page cache / build caches
python -u tools/wino_one.py "$@" 20000 > /tmp/pp.log 2>&1 &
pid=$!
sleep 8
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*: (\([0-9]*Mhz\)).*/sclk \1/; s/.*Power (W): /power W /' | tr '\n' ' '; echo; sleep 0.5; done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
