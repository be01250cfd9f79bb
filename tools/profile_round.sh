# Round profile: GPU tests, smoke, bench line, rocprofv3 kernel stats and separate PMC passes.
# usage (GPU box): bash tools/profile_round.sh ; outputs under gpurun_out/r01b/
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r01b; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2> $O/pmc_sq.err
find $O -name "*.csv" | head -20
cat $O/pytest_gpu.txt $O/smoke.txt; tail -c 600 $O/bench.json
# summaries the judge reads (copy into profiles/ from the container: gpurun_out/ is scratch):
#   python tools/pmc_traffic.py $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv \
#          profiles/r01_pmc_hbm_traffic.json profiles/hbm_traffic.json
#   python tools/pmc_summary.py $O/pmc_sq/p_counter_collection.csv gemm_pp > profiles/r01_pmc_sq_gemm.txt
#   cp $O/bench.json profiles/r01_bench.json; cp $O/stats/bench_kernel_stats.csv profiles/r01_rocprofv3_kernel_stats.csv
