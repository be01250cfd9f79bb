cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "padded" 2>&1 | tail -2
timeout 600 python tools/resample_fuzz.py 120 227 2>&1 | tail -2 | tee -a $O/fuzz_resample_padded.log
AB_BENCH_ARGS="--mode objects --no-cpu-baseline --steps 6 --warmup 2" python tools/ab_env.py 3 dense:OAKE_PADDED_CROPS=0 padded:OAKE_PADDED_CROPS=1 > $O/ab_padded_crops_objects_fullrows.log 2>&1; tail -3 $O/ab_padded_crops_objects_fullrows.log
cd /tmp
for p in 0 1; do
  OAKE_PADDED_CROPS=$p OAKE_BENCH_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_objects_padded$p -o b -- python $GRAFT_REPO_ROOT/bench.py --mode objects --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-profile > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/stats_objects_padded$p -name "*kernel_stats.csv" | head -1)
  echo "== padded=$p"; grep -i "resample\|pad_nchw\|crop_norm" $f | cut -c1-200
  find $GRAFT_REPO_ROOT/$O/stats_objects_padded$p -name "*kernel_trace.csv" -delete; find $GRAFT_REPO_ROOT/$O/stats_objects_padded$p -name "*agent_info.csv" -delete
done
