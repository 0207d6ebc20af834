bash tools/prof_round.sh r05final 7738363 > gpurun_out/prof_final.log 2>&1; tail -2 gpurun_out/prof_final.log | cut -c1-300
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05final/bench20.json 2> gpurun_out/r05final/bench20.err; cut -c1-400 gpurun_out/r05final/bench20.json
