cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_resample_gpu.py tests/test_edge_cases_gpu.py tests/test_oake_gpu.py tests/test_jpeg.py tests/test_encoder_gpu.py -x -q -m gpu -k "not fused_qkv" 2>&1 | tail -3 | tee $O/call9_pytest.txt
for s in 231 233; do timeout 600 python tools/resample_fuzz.py 150 $s 2>&1 | tail -2; done | tee $O/fuzz_resample_quadstore.log
timeout 600 python tools/blocks_fuzz.py 40 231 2>&1 | tail -1 | tee -a $O/fuzz_resample_quadstore.log
timeout 900 python tests/fuzz_pipeline.py 12 231 2>&1 | tail -1 | tee -a $O/fuzz_resample_quadstore.log
cd /tmp
for m in objects blocks; do
  OAKE_BENCH_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o b -- python $GRAFT_REPO_ROOT/bench.py --mode $m --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-profile > /dev/null 2>&1
  f=$(find /tmp/st -name "*kernel_stats.csv" | head -1)
  echo "== $m"; grep -i "resample\|pad_nchw\|crop_norm" $f | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-150
  rm -rf /tmp/st
done 2>&1 | tee $GRAFT_REPO_ROOT/$O/resample_kernel_stats_quadstore.txt
cd $GRAFT_REPO_ROOT
python bench.py --mode objects --no-cpu-baseline --steps 6 --warmup 2 | tail -1 | cut -c1-300
