O=gpurun_out/r05n; mkdir -p $O
L() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', j['value'], j['ms_per_step'])"; }
for r in 1 2 3; do
python3 bench.py --streams 1 --steps 200 --warmup 20 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs 2>/dev/null | L "single full gc-off"
python3 bench.py --streams 1 --steps 200 --warmup 20 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs --keep-gc 2>/dev/null | L "single full gc-on"
python3 bench.py --streams 1 --fic-only --steps 200 --warmup 20 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs 2>/dev/null | L "single fic gc-off"
python3 bench.py --streams 1 --fic-only --steps 200 --warmup 20 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs --keep-gc 2>/dev/null | L "single fic gc-on"
done
for r in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs 2>/dev/null | L "512 gc-off"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs --keep-gc 2>/dev/null | L "512 gc-on"
done
