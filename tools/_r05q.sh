L() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); d=j['config'].get('delivered_to_host') or {}; print('$1', j['value'], d.get('frames_per_s_without_delivery_same_steps'), d.get('frac_of_that'))"; }
B="--steps 98 --warmup 14 --no-cpu-baseline --no-host-leg --no-single-legs"
for r in 1 2 3; do
python3 bench.py $B --deliver 2>/dev/null | L "deliver all "
python3 bench.py $B --deliver --deliver-what 1 2>/dev/null | L "deliver fib "
python3 bench.py $B --deliver --deliver-what 2 2>/dev/null | L "deliver msc "
done
