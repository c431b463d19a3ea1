# GPU box: sample clocks / power while a bench workload runs.  usage: tools/dbg/clock_watch.sh "<bench args>"
python bench.py $1 --no-cpu-baseline > /tmp/cw.json 2>/dev/null &
PID=$!
sleep 6
for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s " \t" " " | tr "\n" "|"; echo; sleep 0.5; kill -0 $PID 2>/dev/null || break; done
wait $PID
python -c "
import json; d=json.load(open('/tmp/cw.json')); r=d['roofline']; print(d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items() if v>0.006})"
