"""Copy the judged artefacts of a tools/profile_round.sh session from gpurun_out/<tag>/ (scratch) into
profiles/<tag>/ (tracked) — raw tool output only, nothing edited — one directory per mode:

    bench.json                                   the driver-contract line of the mode (two lanes)
    lane1_bench.json                             the same line on ONE lane (OAKE_BENCH_LANES=1)
    lane1_rocprofv3_kernel_stats.csv             rocprofv3 --kernel-trace --stats of the one-lane command
    lane1_rocprofv3_kernel_trace.csv.gz          ... and its per-dispatch trace (begin / end of every launch; gzip -9)
    lane1_bench_under_rocprof.json               the line that profiled run printed
    lane1_pmc_{fetch,write,sq}_counter_collection.csv   separate --pmc passes of the one-lane command
    lanes2_rocprofv3_kernel_stats.csv            kernel stats with two lanes overlapping (for the record)

Derived numbers (per-launch HBM traffic, MFMA utilisation, HBM GB/s) are made from these by
tools/derive_counters.py.   usage: python tools/collect_profiles.py [tag=r04]"""
import glob
import pathlib
import shutil
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
src, dst = ROOT / 'gpurun_out' / tag, ROOT / 'profiles' / tag
dst.mkdir(parents=True, exist_ok=True)
for name in ('session.txt', 'pytest_gpu.txt', 'smoke.txt'):
    if (src / name).exists():
        shutil.copy(src / name, dst / name)

for mode_dir in sorted(p for p in src.iterdir() if p.is_dir() and (p / 'bench.json').exists()):
    out = dst / mode_dir.name
    out.mkdir(exist_ok=True)
    shutil.copy(mode_dir / 'bench.json', out / 'bench.json')
    for rel, name in (('lane1/bench.json', 'lane1_bench.json'),
                      ('lane1/bench_under_rocprof.json', 'lane1_bench_under_rocprof.json'),
                      ('lanes2/bench_under_rocprof.json', 'lanes2_bench_under_rocprof.json')):
        if (mode_dir / rel).exists():
            shutil.copy(mode_dir / rel, out / name)
    for pat, name in (('lane1/stats/**/*kernel_stats.csv', 'lane1_rocprofv3_kernel_stats.csv'),
                      ('lane1/stats/**/*kernel_trace.csv.gz', 'lane1_rocprofv3_kernel_trace.csv.gz'),
                      ('lanes2/stats/**/*kernel_stats.csv', 'lanes2_rocprofv3_kernel_stats.csv'),
                      ('lane1/pmc_fetch/**/*counter_collection.csv', 'lane1_pmc_fetch_counter_collection.csv'),
                      ('lane1/pmc_write/**/*counter_collection.csv', 'lane1_pmc_write_counter_collection.csv'),
                      ('lane1/pmc_sq/**/*counter_collection.csv', 'lane1_pmc_sq_counter_collection.csv')):
        hits = glob.glob(str(mode_dir / pat), recursive=True)
        if hits:
            shutil.copy(hits[0], out / name)
    print(mode_dir.name, sorted(p.name for p in out.iterdir()))
print((dst / 'session.txt').read_text().splitlines()[0] if (dst / 'session.txt').exists() else 'no session stamp', '->', dst)
