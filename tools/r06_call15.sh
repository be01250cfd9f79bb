# round 6, GPU call 15: the 320 x 256 tile without DMA waves (gemm_w8_kernel, lab variant 13): parity, launch times, in-situ A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/w8; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_16bit_epilogues or gemm_layernorm_folded" 2>&1 | tail -5 | tee $O/pytest.txt
timeout 300 python tools/gemm_ablate.py 4,13 5 25600 2>&1 | grep -v amdgpu.ids | tee $O/gemm_ablate_w8_m25600.txt
for r in 1 2; do for v in 4 13; do
  OAKE_LAB_LIB=1 OAKE_GEMM_VARIANT=$v timeout 300 python bench.py --mode globals 2>/dev/null | tail -1 > $O/bench_globals_v${v}_r$r.json
  python - <<PY
import json
d = json.load(open('$O/bench_globals_v${v}_r$r.json'))
print('variant $v run $r', d['value'], d['roofline']['frac'], d['roofline'].get('achieved'))
PY
done; done 2>&1 | tee $O/ab.txt
