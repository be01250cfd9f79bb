# K-loop ablation of the persistent GEMM on the final binary — GPU box.
# Builds (made in the container, they travel with the snapshot):
#   for n in 1 2 3 4; do OAKE_LIB_OUT=oadp_amd/liboake_abl$n.so OAKE_EXTRA_FLAGS="-DOAKE_LAB=1 -DOAKE_KLOOP_ABLATE=$n" python -m oadp_amd.build; done
# (bit 1 = half of the fragment reads, 2 = half of the LDS-DMA pieces, 4 = half of the MFMAs; results are WRONG by design).
# Per build: cycle stamps of the four-phase form on c_fc's and c_proj's shapes (tools/gemm_trace.py, variant 10), the K loop
# alone ('none' epilogue) in four phases and in the production two long phases (tools/gemm_ablate.py 4,9), and the one-lane
# bench with its live per-kernel table (the production kernels in situ: c_fc = EPI_T16_GELU_LN in long phases).
# usage: bash tools/kloop_ablate.sh [outdir=gpurun_out/r06/kloop]
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r06/kloop}; mkdir -p $O
for n in 0 1 2 3 4; do
  if [ $n = 0 ]; then L=oadp_amd/liboake_hip_lab.so; else L=oadp_amd/liboake_abl$n.so; fi
  [ -f $L ] || { echo "missing $L"; continue; }
  echo "=== ablate $n ($L)" | tee -a $O/summary.txt
  for shape in "12800 3072 768 bias" "12800 768 3072 resid" "12800 768 768 resid"; do
    OAKE_LAB_LIB=$L timeout 120 python tools/gemm_trace.py $shape 10 2>&1 | grep -v Warning | tee -a $O/trace_abl$n.txt | head -4 | tee -a $O/summary.txt
  done
  OAKE_LAB_LIB=$L timeout 300 python tools/gemm_ablate.py 4,9 3 2>&1 | grep -v Warning | tee $O/none_abl$n.txt | tee -a $O/summary.txt
  OAKE_LIB=$L OAKE_BENCH_LANES=1 OAKE_BENCH_FULL_LINE=1 OAKE_BENCH_SKIP_FINITE=1 timeout 300 python bench.py --no-modes --no-cpu-baseline --steps 40 > $O/bench_abl$n.json 2> $O/bench_abl$n.err
  python - $O/bench_abl$n.json <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    k = d['kernels']
    print('bench one lane: %.1f images/s;' % d['value'], '  '.join(f"{n} {1e3*v['ms_per_step']/max(v['launches_per_step'],1):.1f}us" for n, v in list(k.items())[:7]))
except Exception as e:
    print('bench failed', e)
PY
done
