"""HBM-side traffic of the persistent GEMM by tile order — run UNDER rocprofv3 --pmc FETCH_SIZE (and again
--pmc WRITE_SIZE):  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- python tools/panel_traffic.py run
then:  python tools/panel_traffic.py report out/**/p_counter_collection.csv
Each (shape, panel) is REPS consecutive launches of the c_fc / qkv epilogue GEMM; the report groups the
counter rows of the GEMM kernel in dispatch order.  Panels: n > 0 = N panels of n tile columns (row-major
inside), n < 0 = M slabs of -n tile rows."""
import sys, os
SHAPES = [('c_fc', 12800, 3072, 768, 1), ('qkv', 12800, 2304, 768, 0)]
PANELS = [3, 4, 6, 12, -5, -10]
REPS = 6
if sys.argv[1] == 'run':
    import ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from oadp_amd import _lib
    lib = _lib.load_lab()  # the build that carries every variant
    dev = torch.device('cuda:0')
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, m, n, k, gelu in SHAPES:
        a = (torch.randn(m, k, device=dev) * 0.5).half()
        w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
        bias = torch.randn(n, device=dev)
        c = torch.empty(m, n, device=dev, dtype=torch.float16)
        for p in PANELS:
            lib.oake_debug_set_gemm_panel(p)
            for _ in range(REPS):
                assert lib.oake_debug_gemm16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, gelu, s) == 0
            torch.cuda.synchronize()
    lib.oake_debug_set_gemm_panel(0)
else:
    import csv
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            if 'gemm_pp_kernel' in r['Kernel_Name']:
                rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value'])))
    rows.sort()
    vals = [v for _, _, v in rows]
    cname = rows[0][1] if rows else '?'
    i = 0
    for name, m, n, k, gelu in SHAPES:
        alg = (m * k + n * k) * 2 if cname == 'FETCH_SIZE' else m * n * 2
        for p in PANELS:
            grp = vals[i:i + REPS][1:]  # skip the first launch of a group (cold)
            i += REPS
            mean_kib = sum(grp) / max(len(grp), 1)
            mb = mean_kib * 1024 / 1e6 * (2 if cname == 'FETCH_SIZE' else 1)
            print(f'{name:5s} panel {p:4d}: {cname} {mb:8.1f} MB/launch (x{mb * 1e6 / alg:5.2f} of the algorithmic {alg / 1e6:.1f} MB)')
