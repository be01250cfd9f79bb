# Round profile session (GPU box): for each bench mode the driver-contract line (two lanes, as the driver runs it),
# then — on ONE lane (OAKE_BENCH_LANES=1: every kernel strictly after its predecessor, which is the condition
# bench.py's own per-kernel stamps are taken under) — the rocprofv3 --kernel-trace --stats summary of the same
# command and separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ); plus the two-lane kernel stats for the record.
# All RAW tool output, under gpurun_out/<tag>/<mode>/, with a session stamp every derived number carries.
# usage: bash tools/profile_round.sh [tag=r06] [modes="globals blocks objects blocks_1700x1134"] [tests=1]
# (blocks_1700x1134 = --mode blocks --image-size 1700x1134: BASELINE.md 4's 5-level configuration, 245 crops per image)
# In the container afterwards:  python tools/collect_profiles.py <tag> && python tools/derive_counters.py <tag>
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06}
MODES=${2:-"globals blocks objects blocks_1700x1134"}
export OAKE_BENCH_FULL_LINE=1  # bench.json files under profiles/ are the full records (stdout of a plain run is the compact line)
TESTS=${3:-1}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
SESSION="$(hostname)-$(date -u +%Y%m%dT%H%M%SZ)"
{ echo "session: $SESSION"; echo "head: $(cat .git/HEAD 2>/dev/null || echo n/a)"; rocm-smi --showproductname 2>/dev/null | grep -i -m2 "card series\|gfx" ; rocm-smi --showmaxpower --showpower 2>/dev/null | grep -i "power" ; } > $O/session.txt
if [ "$TESTS" = "1" ]; then
  env -u OAKE_BENCH_FULL_LINE python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
fi
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"
for M in $MODES; do
  D=$O/$M; mkdir -p $D/lane1 $D/lanes2
  B="python $GRAFT_REPO_ROOT/bench.py --mode $M"
  if [ "$M" = "blocks_1700x1134" ]; then B="python $GRAFT_REPO_ROOT/bench.py --mode blocks --image-size 1700x1134 --steps 8 --warmup 2"; fi
  (cd $GRAFT_REPO_ROOT && $B > $D/bench.json 2> $D/bench.err)
  (cd $GRAFT_REPO_ROOT && OAKE_BENCH_LANES=1 $B --no-cpu-baseline --no-modes > $D/lane1/bench.json 2> $D/lane1/bench.err)
  cd /tmp
  Q="--no-cpu-baseline --no-modes"
  # the stats passes run >= 1 s of steps: the shader clock takes ~30 ms of load to ramp (profiles/r04/clock_ramp.txt)
  # and rocprofv3's AverageNs is over ALL calls of the run
  case $M in globals) K="--steps 400 --warmup 50";; blocks) K="--steps 60 --warmup 10";; blocks_1700x1134) K="--steps 8 --warmup 2";; *) K="--steps 12 --warmup 3";; esac
  OAKE_BENCH_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D/lane1/stats -o bench -- $B $K $Q > $D/lane1/bench_under_rocprof.json 2> $D/lane1/rocprof_stats.err
  OAKE_BENCH_LANES=1 OAKE_BENCH_RAMP_S=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/lane1/pmc_fetch -o p -- $B --steps 2 --warmup 1 $Q --no-profile > /dev/null 2> $D/lane1/pmc_fetch.err
  OAKE_BENCH_LANES=1 OAKE_BENCH_RAMP_S=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/lane1/pmc_write -o p -- $B --steps 2 --warmup 1 $Q --no-profile > /dev/null 2> $D/lane1/pmc_write.err
  OAKE_BENCH_LANES=1 OAKE_BENCH_RAMP_S=0 rocprofv3 --pmc $SQ --output-format csv -d $D/lane1/pmc_sq -o p -- $B --steps 2 --warmup 1 $Q --no-profile > /dev/null 2> $D/lane1/pmc_sq.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/lanes2/stats -o bench -- $B $K $Q > $D/lanes2/bench_under_rocprof.json 2> $D/lanes2/rocprof_stats.err
  # keep what is judged, drop the bulky per-dispatch traces
  find $D/lanes2 -name "*kernel_trace.csv" -delete; find $D -name "*agent_info.csv" -delete
  find $D/lane1 -name "*kernel_trace.csv" -exec gzip -9 {} \;  # (lane1's trace stays, compressed: per-dispatch durations)
  cd $GRAFT_REPO_ROOT
done
find $O -type f | xargs ls -la | awk '{print $5, $9}' | sort -n | tail -40
cat $O/session.txt $O/pytest_gpu.txt $O/smoke.txt
for M in $MODES; do tail -c 300 $O/$M/bench.json; echo; done
