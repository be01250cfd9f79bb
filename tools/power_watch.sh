# Sample power / clocks with rocm-smi while a command runs: bash tools/power_watch.sh <cmd...>
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' ; echo; sleep 0.2; done ) > gpurun_out/power_watch.log 2>&1 &
W=$!
"$@"
kill $W
sed -E 's/GPU\[0\]\s*: //g; s/\s+/ /g' gpurun_out/power_watch.log | awk 'NR%3==0' | head -40
