"""Copy the judged artefacts of a tools/profile_round.sh session from gpurun_out/<tag>/ (scratch) into
profiles/<tag>/ (tracked) — raw tool output only — and derive, beside them, the per-launch HBM traffic
table bench.py quotes (clearly marked as derived, with the session it came from).
usage: python tools/collect_profiles.py [tag=r03]"""
import collections
import csv
import glob
import json
import os
import pathlib
import shutil
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
src, dst = ROOT / 'gpurun_out' / tag, ROOT / 'profiles' / tag
dst.mkdir(parents=True, exist_ok=True)
session = 'unknown'
for name in ('session.txt', 'pytest_gpu.txt', 'smoke.txt'):
    if (src / name).exists():
        shutil.copy(src / name, dst / name)
if (src / 'session.txt').exists():
    session = (src / 'session.txt').read_text().splitlines()[0].split(':', 1)[1].strip()

# profiler slot name <- kernel-name fragment (the residual GEMM instantiation serves out_proj and c_proj)
NAMES = {'im2col_kernel': 'im2col', 'pad_nchw_kernel': 'pad_nchw', 'gemm_pp_kernelIDF16_Li6E': 'gemm_conv1', 'gemm_pp_kernelIDF16_Li7E': 'gemm_qkv',
         'gemm_pp_kernelIDF16_Li8E': 'gemm_c_fc', 'gemm_pp_kernelIDF16_Li5E': 'gemm_resid16(out_proj+c_proj)',
         'attention_pair_kernelIDF16_': 'attention', 'attention_coop_kernelIDF16_': 'attention',
         'embed_ln_pre_kernel': 'embed_ln_pre', 'crop_normalize_jobs_kernel': 'crop_normalize',
         'resample_h_kernel': 'resample_h', 'resample_v_kernel': 'resample_v'}


# algorithmic bytes per launch of the kernel bench.py's `roofline` object names: A + W + output once
# globals c_fc: M 12800 (256 crops x 50 tokens), N 3072, K 768, 16-bit: 19.66 + 4.72 + 78.64 MB
ALGORITHMIC = {('globals', 'gemm_c_fc'): 2 * (12800 * 768 + 3072 * 768 + 12800 * 3072)}


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


for mode_dir in sorted(p for p in src.iterdir() if p.is_dir()):
    mode = mode_dir.name
    out = dst / mode
    out.mkdir(exist_ok=True)
    for name in ('bench.json', 'bench_under_rocprof.json'):
        if (mode_dir / name).exists():
            shutil.copy(mode_dir / name, out / name)
    for pat, name in (('stats/**/*kernel_stats.csv', 'rocprofv3_kernel_stats.csv'),
                      ('pmc_fetch/**/*counter_collection.csv', 'pmc_fetch_counter_collection.csv'),
                      ('pmc_write/**/*counter_collection.csv', 'pmc_write_counter_collection.csv'),
                      ('pmc_sq/**/*counter_collection.csv', 'pmc_sq_counter_collection.csv')):
        hits = glob.glob(str(mode_dir / pat), recursive=True)
        if hits:
            shutil.copy(hits[0], out / name)
    fp, wp = out / 'pmc_fetch_counter_collection.csv', out / 'pmc_write_counter_collection.csv'
    if not (fp.exists() and wp.exists()):
        continue
    fetch, write = per_kernel(fp, 'FETCH_SIZE'), per_kernel(wp, 'WRITE_SIZE')
    table = {'_session': session, '_derived_from': [str(fp.relative_to(ROOT)), str(wp.relative_to(ROOT))],
             '_note': 'DERIVED by tools/collect_profiles.py: mean per launch; FETCH_SIZE (KiB) doubled per '
                      'MI355X_MICROARCH.md (gfx950 counts a wide coalesced read stream at half its bytes), '
                      'WRITE_SIZE (KiB) as is'}
    for k, (f_kib, n) in fetch.items():
        if 'oake' not in k:
            continue
        w_kib = write.get(k, (0.0, 0))[0]
        rec = {'kernel': k, 'launches_sampled': n, 'fetch_kib_raw': round(f_kib, 1), 'write_kib_raw': round(w_kib, 1),
               'hbm_read_bytes_corrected': int(2 * f_kib * 1024), 'hbm_write_bytes': int(w_kib * 1024),
               'hbm_bytes_per_launch': int((2 * f_kib + w_kib) * 1024)}
        for frag, slot in NAMES.items():
            if frag in k:
                if (mode, slot) in ALGORITHMIC:
                    rec['algorithmic_bytes_per_launch'] = ALGORITHMIC[(mode, slot)]
                table[slot] = rec
    tagname = '' if mode == 'globals' else f'{mode}_'
    (ROOT / 'profiles' / f'{tag}_{tagname}hbm_traffic.json').write_text(json.dumps(table, indent=1))
    stats = out / 'rocprofv3_kernel_stats.csv'
    if stats.exists():
        shutil.copy(stats, ROOT / 'profiles' / f'{tag}_{tagname}rocprofv3_kernel_stats.csv')
    print(mode, {k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in table.items() if not k.startswith('_')})
print('session', session, '->', dst)
