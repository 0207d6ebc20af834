O=gpurun_out/r05y; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench.log 2>&1
cd $GRAFT_REPO_ROOT; grep -E "k_ingest|k_deliver|k_msc_vitT|k_convert" $O/stats/*/*kernel_stats.csv | cut -c1-220
