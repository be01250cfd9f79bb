# Round profile session (GPU box): for each bench mode the driver-contract line, the rocprofv3
# --kernel-trace --stats summary of the same command and separate PMC passes (FETCH_SIZE / WRITE_SIZE /
# SQ), all RAW tool output, under gpurun_out/<tag>/<mode>/ ; plus a session stamp every derived number
# carries.  usage: bash tools/profile_round.sh [tag=r03] [modes="globals blocks objects"]
# Copy what is to be judged into profiles/<tag>/ afterwards (gpurun_out/ is scratch), then derive the
# per-launch HBM traffic with tools/pmc_traffic.py (see the end of this file).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}
MODES=${2:-"globals blocks objects"}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
SESSION="$(hostname)-$(date -u +%Y%m%dT%H%M%SZ)"
{ echo "session: $SESSION"; echo "head: $(cat .git/HEAD 2>/dev/null || echo n/a)"; rocm-smi --showproductname 2>/dev/null | grep -i -m2 "card series\|gfx" ; rocm-smi --showmaxpower --showpower 2>/dev/null | grep -i "power" ; } > $O/session.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
for M in $MODES; do
  D=$O/$M; mkdir -p $D
  B="python $GRAFT_REPO_ROOT/bench.py --mode $M"
  (cd $GRAFT_REPO_ROOT && $B > $D/bench.json 2> $D/bench.err)
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o bench -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-modes > $D/bench_under_rocprof.json 2> $D/rocprof_stats.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o p -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-modes > /dev/null 2> $D/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o p -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-modes > /dev/null 2> $D/pmc_write.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $D/pmc_sq -o p -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-modes > /dev/null 2> $D/pmc_sq.err
  # keep what is judged, drop the bulky per-dispatch traces
  find $D -name "*kernel_trace.csv" -delete; find $D -name "*agent_info.csv" -delete
  cd $GRAFT_REPO_ROOT
done
find $O -type f | xargs ls -la | awk '{print $5, $9}' | sort -n | tail -40
cat $O/session.txt $O/pytest_gpu.txt $O/smoke.txt
for M in $MODES; do tail -c 300 $O/$M/bench.json; echo; done
# In the container afterwards:
#   for M in globals blocks objects; do mkdir -p profiles/r02/$M; cp gpurun_out/r02/$M/bench.json gpurun_out/r02/$M/bench_under_rocprof.json profiles/r02/$M/;
#     cp gpurun_out/r02/$M/stats/*/bench_kernel_stats.csv profiles/r02/$M/rocprofv3_kernel_stats.csv; ... (tools/collect_profiles.py)
