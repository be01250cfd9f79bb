# round 6, GPU call 3: what the LDS-DMA stream costs in POWER — form 8 (every piece issued, odd pieces re-fetch their even
# neighbour's lines) beside form 2 (odd pieces dropped) and the base, one-lane bench in situ + cycle stamps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/kloop; mkdir -p $O
for n in 0 8 2 0 8 2; do
  if [ $n = 0 ]; then L=oadp_amd/liboake_hip_lab.so; else L=oadp_amd/liboake_abl$n.so; fi
  OAKE_LIB=$L OAKE_BENCH_LANES=1 OAKE_BENCH_FULL_LINE=1 OAKE_BENCH_SKIP_FINITE=1 timeout 300 python bench.py --no-modes --no-cpu-baseline --steps 40 > $O/bench2_abl$n.json 2> $O/bench2_abl$n.err
  python - $O/bench2_abl$n.json $n <<'PY' | tee -a $O/summary_power.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
k = d['kernels']; s = d.get('sustained') or {}
print('form', sys.argv[2], 'one lane %.1f images/s;' % d['value'], 'sustained', s.get('value'), s.get('power_w'), 'W', s.get('sclk_mhz'), 'MHz;',
      '  '.join(f"{n} {1e3*v['ms_per_step']/max(v['launches_per_step'],1):.1f}us" for n, v in list(k.items())[:5]))
PY
done
OAKE_LAB_LIB=oadp_amd/liboake_abl8.so timeout 120 python tools/gemm_trace.py 12800 768 3072 resid 10 2>&1 | grep -v amdgpu.ids | head -3 | tee -a $O/summary_power.txt
