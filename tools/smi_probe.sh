#!/bin/bash
# What clock / power telemetry the GPU box offers (for bench.py's side-thread sampler)
for d in /sys/class/drm/card*/device; do
  echo "== $d"; ls $d | tr '\n' ' ' | head -c 1500; echo
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo
    for f in power1_average power1_input power1_cap freq1_input freq2_input temp1_input; do [ -r $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done
  done
  for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent current_link_speed; do [ -r $d/$f ] && { echo "$f:"; cat $d/$f 2>&1 | head -12; }; done
done
python -c "import amdsmi; print('amdsmi importable', amdsmi.__file__)" 2>&1 | tail -1
which rocm-smi amd-smi
( time rocm-smi --showpower --showclocks --json ) 2>&1 | tail -12
( time amd-smi metric --power --clock --json ) 2>&1 | head -60
