# round 6, GPU call 18: gemm_w8_kernel with the residual epilogue (out_proj / c_proj at 25 600 rows per pass: blocks and
# objects modes) — GEMM tests incl. the bit-for-bit comparison, then per mode the automatic choice with (-1) and without (-2)
# the 320-row tile, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/w8; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -3 | tee $O/pytest_gemm_resid.txt
for r in 1 2; do for m in blocks objects globals; do for v in -2 -1; do
  OAKE_GEMM_VARIANT=$v OAKE_BENCH_FULL_LINE=1 timeout 600 python bench.py --mode $m --no-cpu-baseline 2>/dev/null | tail -1 > $O/ab_${m}_v${v}_r$r.json
  python - <<PY
import json
d = json.load(open('$O/ab_${m}_v${v}_r$r.json'))
k = d.get('kernels', {})
print('$m variant $v run $r', d['value'], d['unit'], ' '.join(f"{n} {k[n]['ms_per_step']:.3f}" for n in ('gemm_c_fc', 'gemm_c_proj', 'gemm_out_proj', 'qkv_attn') if n in k))
PY
done; done; done 2>&1 | tee $O/ab_modes.txt
