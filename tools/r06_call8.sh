cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_resample_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu -k "padded or resample or crop or edge" 2>&1 | tail -2
timeout 600 python tools/resample_fuzz.py 120 229 2>&1 | tail -2 | tee -a $O/fuzz_resample_padded.log
cd /tmp
for cfg in "1 oadp_amd/liboake_hip.so" "0 oadp_amd/liboake_hip.so" "1 oadp_amd/liboake_tail0.so" "1 oadp_amd/liboake_hip.so" "1 oadp_amd/liboake_tail0.so"; do
  set -- $cfg
  OAKE_LIB=$GRAFT_REPO_ROOT/$2 OAKE_PADDED_CROPS=$1 OAKE_BENCH_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o b -- python $GRAFT_REPO_ROOT/bench.py --mode objects --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-profile > /dev/null 2>&1
  f=$(find /tmp/st -name "*kernel_stats.csv" | head -1)
  echo "== padded=$1 lib=$2"; grep -i "resample\|pad_nchw" $f | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-150
  rm -rf /tmp/st
done 2>&1 | tee $GRAFT_REPO_ROOT/$O/resample_kernel_stats_ab.txt
cd $GRAFT_REPO_ROOT
AB_BENCH_ARGS="--mode objects --no-cpu-baseline --steps 6 --warmup 2" python tools/ab_env.py 3 dense:OAKE_PADDED_CROPS=0 padded:OAKE_PADDED_CROPS=1 > $O/ab_padded_crops_objects_v3.log 2>&1; tail -3 $O/ab_padded_crops_objects_v3.log
