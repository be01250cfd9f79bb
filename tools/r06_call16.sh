# round 6, GPU call 16: gemm_w8_kernel (lab variant 13) in situ — bench globals with every 16-bit tile epilogue (c_fc) on the
# 320 x 256 kernel against the production choice, both from the lab build, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/w8; mkdir -p $O
for r in 1 2 3; do for v in 4 13; do
  OAKE_LIB=oadp_amd/liboake_hip_lab.so OAKE_GEMM_VARIANT=$v timeout 300 python bench.py --mode globals 2>$O/bench_v${v}_r$r.err | tail -1 > $O/bench_globals_v${v}_r$r.json
  python - <<PY
import json
try:
    d = json.load(open('$O/bench_globals_v${v}_r$r.json'))
    print('variant $v run $r', d['value'], d['roofline']['frac'], d['roofline'].get('achieved'), d['roofline'].get('kernel'))
except Exception as e:
    print('variant $v run $r failed', e); print(open('$O/bench_v${v}_r$r.err').read()[-1500:])
PY
done; done 2>&1 | tee $O/ab.txt
