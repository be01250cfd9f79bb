# round 6, GPU call 13: c_fc's N-panel width once more (second half) now that the tile stores are written
# through: bench A/B + FETCH_SIZE of gemm_c_fc per panel width
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
python tools/ab_env.py 3 pn3:OAKE_GEMM_PANEL=0 pn4:OAKE_GEMM_PANEL=4 pn6:OAKE_GEMM_PANEL=6 > $O/ab_gemm_panel_globals.log 2>&1; tail -4 $O/ab_gemm_panel_globals.log
cd /tmp
for pn in 0 4 6; do
  OAKE_GEMM_PANEL=$pn OAKE_BENCH_LANES=1 OAKE_BENCH_RAMP_S=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf$pn -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes --no-profile > /dev/null 2>&1
  python - /tmp/pf$pn $pn <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE': acc[r['Kernel_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    if 'Li8E' in k and 'gemm_pp' in k: print('panel', sys.argv[2], 'gemm_c_fc reads %.1f MB/launch' % (2 * 1024 * sum(v) / len(v) / 1e6), 'n', len(v))
    if 'Li7E' in k and 'gemm_pp' in k: print('panel', sys.argv[2], 'gemm_kv reads %.1f MB/launch' % (2 * 1024 * sum(v) / len(v) / 1e6), 'n', len(v))
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/$O/gemm_panel_fetch.txt
