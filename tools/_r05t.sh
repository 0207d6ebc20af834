BENCH_ARGS="--steps 63 --warmup 18 --no-deliver-leg --no-host-leg --no-single-legs" bash tools/ab.sh gpurun_out/r05t/ab 3 "ber13|dabstar_amd/_ab/libdabx_ber13.so||" "ber4|dabstar_amd/_ab/libdabx_ber4.so||" "berloop|dabstar_amd/_ab/libdabx_berloop.so||"
L() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', j['value'], j.get('frames_per_s_per_locked_stream'), j['host_us_per_step'])"; }
B="--steps 20 --warmup 5 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs"
for r in 1 2 3; do
DABX_LIB=$PWD/dabstar_amd/_ab/libdabx_berloop.so python3 bench.py $B --sync-calls 2>/dev/null | L "sync base    "
DABX_LIB=$PWD/dabstar_amd/_ab/libdabx_berloop.so python3 bench.py $B --sync-calls --unlocked 8 2>/dev/null | L "sync unlocked"
DABX_LIB=$PWD/dabstar_amd/_ab/libdabx_berloop.so python3 bench.py $B 2>/dev/null | L "async base   "
DABX_LIB=$PWD/dabstar_amd/_ab/libdabx_berloop.so python3 bench.py $B --unlocked 8 2>/dev/null | L "async unlocked"
done
