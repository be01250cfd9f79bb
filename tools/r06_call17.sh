# round 6, GPU call 17: gemm_w8_kernel in the production library (automatic choice: c_fc) — kernel + encoder suites, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/w8; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -3 | tee $O/pytest_gemm.txt
timeout 1500 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_encoder.txt
for r in 1 2; do for v in 4 -1; do
  OAKE_GEMM_VARIANT=$v timeout 300 python bench.py --mode globals 2>/dev/null | tail -1 > $O/prod_globals_v${v}_r$r.json
  python - <<PY
import json
d = json.load(open('$O/prod_globals_v${v}_r$r.json'))
print('production library, variant $v run $r', d['value'], d['roofline']['frac'], d['roofline'].get('achieved'), d['roofline'].get('avg_launch_us'))
PY
done; done 2>&1 | tee $O/ab_prod.txt
