cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_resample_gpu.py tests/test_edge_cases_gpu.py tests/test_oake_gpu.py tests/test_oake_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/call11_pytest.txt
for s in 241 243; do timeout 600 python tools/resample_fuzz.py 150 $s 2>&1 | tail -1; timeout 600 python tools/blocks_fuzz.py 50 $s 2>&1 | tail -1; done | tee $O/fuzz_frontend_v3.log
timeout 900 python tests/fuzz_pipeline.py 12 241 2>&1 | grep -v "^\[" | tail -3 | tee -a $O/fuzz_frontend_v3.log
cd /tmp
OAKE_BENCH_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o b -- python $GRAFT_REPO_ROOT/bench.py --mode blocks --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-profile > /dev/null 2>&1
f=$(find /tmp/st -name "*kernel_stats.csv" | head -1)
grep -i "resample\|crop_norm" $f | awk -F'",' '{print substr($1,1,70), $2}' | cut -c1-150 | tee $GRAFT_REPO_ROOT/$O/blocks_frontend_kernel_stats.txt
cd $GRAFT_REPO_ROOT
python bench.py --mode blocks --no-cpu-baseline --steps 20 --warmup 4 | tail -1 | cut -c1-200
python bench.py --mode blocks --image-size 1700x1134 --no-cpu-baseline --steps 6 --warmup 2 | tail -1 | cut -c1-200
