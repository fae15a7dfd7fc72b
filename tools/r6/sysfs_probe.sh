#!/bin/bash
# what does the GPU box expose for clocks / power? (round-6 calibration block of bench.py)
for c in /sys/class/drm/card*/device; do
  echo "== $c"; cat $c/vendor $c/device 2>/dev/null | tr '\n' ' '; echo
  ls $c | tr '\n' ' ' | cut -c1-1500; echo
  for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent current_link_speed; do [ -e $c/$f ] && { echo "-- $f"; cat $c/$f 2>&1 | head -12; }; done
  for h in $c/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in freq1_input freq2_input power1_average power1_input power1_cap temp1_input; do [ -e $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done; done
done
echo "== rocm-smi"; time rocm-smi --showclocks --showpower --json 2>&1 | head -30
time amd-smi metric --clock --power --json 2>&1 | head -60
nproc; lscpu | head -20
