cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
python tools/ab_env.py 5 pn3:OAKE_GEMM_PANEL=0 pn4:OAKE_GEMM_PANEL=4 pn6:OAKE_GEMM_PANEL=6 > $O/ab_gemm_panel_globals_5r.log 2>&1; tail -3 $O/ab_gemm_panel_globals_5r.log
AB_BENCH_ARGS="--mode objects --no-cpu-baseline --steps 6 --warmup 2" python tools/ab_env.py 3 pn3:OAKE_GEMM_PANEL=0 pn4:OAKE_GEMM_PANEL=4 pn6:OAKE_GEMM_PANEL=6 > $O/ab_gemm_panel_objects.log 2>&1; tail -3 $O/ab_gemm_panel_objects.log
AB_BENCH_ARGS="--mode blocks --no-cpu-baseline --steps 20 --warmup 4" python tools/ab_env.py 3 pn3:OAKE_GEMM_PANEL=0 pn4:OAKE_GEMM_PANEL=4 pn6:OAKE_GEMM_PANEL=6 > $O/ab_gemm_panel_blocks.log 2>&1; tail -3 $O/ab_gemm_panel_blocks.log
