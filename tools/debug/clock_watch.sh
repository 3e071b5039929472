# sample shader clock / power while a workload runs:  bash tools/debug/clock_watch.sh <python args...>
python3 "$@" > /tmp/cw.log 2>&1 &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '; echo; sleep 1; done
wait $PID
tail -1 /tmp/cw.log
