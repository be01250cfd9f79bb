"""DERIVE, from the raw one-lane passes tools/collect_profiles.py copied into profiles/<tag>/<mode>/, the table
north_star asks for — per kernel: rocprofv3 launch duration, MFMA utilisation and HBM GB/s against chip peak —
as profiles/<tag>_mfma_util_hbm.json (what bench.py quotes as roofline.traffic / avg_launch_us_rocprofv3).

  duration   mean of the launch's begin -> end in lane1_rocprofv3_kernel_trace.csv.gz (one lane: no overlap) over
             the second half of the run's launches (the shader clock ramps for ~30 ms after idle:
             profiles/r04/clock_ramp.txt); the mean over all calls — what rocprofv3's stats file prints — beside it
  MFMA       SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per v_mfma_f32_16x16x32 (16 384 FLOP), summed over the chip's 1024
             SIMDs.  util_at_clock = busy / (1024 x GRBM_GUI_ACTIVE / 8)  [GUI_ACTIVE is summed over the 8 XCDs];
             frac_of_peak = issued MFMA FLOP / duration / 2.5 PFLOP/s  (issued >= algorithmic: tile padding counts)
  HBM        FETCH_SIZE (KiB) x 2 [gfx950 correction, MI355X_MICROARCH.md] + WRITE_SIZE (KiB), mean per launch,
             / duration, against 8 TB/s.  Fabric-side counters: Infinity-Cache hits are included.
The residual GEMM instantiation serves out_proj and c_proj alternately (launch order splits them; the SQ pass's
MFMA count confirms the split).   usage: python tools/derive_counters.py [tag=r04]"""
import collections
import csv
import gzip
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
PEAK_FLOPS, PEAK_HBM, SIMDS, XCDS = 2.5e15, 8e12, 1024, 8
SLOTS = {'im2col_kernel': 'im2col', 'pad_nchw_kernel': 'pad_nchw', 'embed_ln_pre_kernel': 'embed_ln_pre',
         'gemm_pp_kernelIDF16_Li6E': 'gemm_conv1', 'gemm_pp_kernelIDF16_Li7E': 'gemm_qkv', 'gemm_pp_kernelIDF16_Li8E': 'gemm_c_fc', 'gemm_w8_kernelIDF16_Li8E': 'gemm_c_fc', 'gemm_w8_kernelIDF16_Li7E': 'gemm_qkv',
         'gemm_w8_kernelIDF16_Li5E': ('gemm_out_proj', 'gemm_c_proj'),  # (the 320-row tile: passes of 25 600 rows)
         'gemm_pp_kernelIDF16_Li5E': ('gemm_out_proj', 'gemm_c_proj'), 'attention_pair_kernel': 'attention',
         'attention_coop_kernel': 'attention', 'qkv_attn_kernel': 'qkv_attn', 'qkv_attn_obj_kernel': 'qkv_attn',
         # objects mode, 12 layers: 11 launches with the patch stream, then the last layer's object token alone
         'attention_head_kernel': ('attention',) * 11 + ('object_attention',), 'attn_out_kernel': 'attn_out', 'object_attention_kernel': 'object_attention',
         'crop_normalize_jobs_kernel': 'crop_normalize', 'resample_h_kernel': 'resample_h', 'resample_v4_kernel': 'resample_v',
         'resample_v_kernel': 'resample_v', 'resample_v4p_kernel': 'resample_v', 'resample_v4_u8_kernel': 'resample_v_u8',
         'resample_coeffs_kernel': 'resample_coeffs'}


def slot_of(name, seen):
    for frag, slot in SLOTS.items():
        if frag in name:
            if isinstance(slot, tuple):  # alternating launches of one instantiation
                seen[frag] += 1
                return slot[(seen[frag] - 1) % len(slot)]
            return slot
    return None


def dispatches(path, by='Dispatch_Id'):
    rows = collections.OrderedDict()
    with (gzip.open(path, 'rt') if str(path).endswith('.gz') else open(path)) as f:
        for r in csv.DictReader(f):
            d = rows.setdefault(int(r[by]), {'name': r['Kernel_Name'], 't0': int(r['Start_Timestamp']), 't1': int(r['End_Timestamp'])})
            if 'Counter_Name' in r:
                d[r['Counter_Name']] = float(r['Counter_Value'])
    return [rows[k] for k in sorted(rows)]


def per_slot(path):
    seen, acc = collections.Counter(), collections.defaultdict(list)
    ds = dispatches(path)
    # objects mode with the fused ln_1 + in_proj + attention kernel: attention_head_kernel is left with the last layer's
    # object token alone (one launch per pass)
    if any('qkv_attn_obj_kernel' in d['name'] for d in ds):
        SLOTS['attention_head_kernel'] = 'object_attention'
    for d in ds:
        s = slot_of(d['name'], seen)
        if s:
            acc[s].append(d)
    return acc


def mean(v):
    return sum(v) / len(v)


out = {'_derived_by': 'tools/derive_counters.py', '_peak': {'mfma_flops': PEAK_FLOPS, 'hbm_bytes_per_s': PEAK_HBM},
       '_session': (ROOT / 'profiles' / tag / 'session.txt').read_text().splitlines()[0].split(':', 1)[1].strip()}
for mode_dir in sorted(p for p in (ROOT / 'profiles' / tag).iterdir() if (p / 'lane1_rocprofv3_kernel_trace.csv.gz').exists()):
    f = {k: mode_dir / f'lane1_{k}' for k in ('rocprofv3_kernel_trace.csv.gz', 'pmc_fetch_counter_collection.csv',
                                               'pmc_write_counter_collection.csv', 'pmc_sq_counter_collection.csv')}
    trace, fetch, write, sq = (per_slot(p) if p.exists() else {} for p in f.values())
    table = {'_derived_from': [str(p.relative_to(ROOT)) for p in f.values() if p.exists()]}
    for slot, ds in trace.items():
        us_all = mean([d['t1'] - d['t0'] for d in ds]) / 1e3
        us = mean([d['t1'] - d['t0'] for d in ds[len(ds) // 2:]]) / 1e3  # second half of the run: clocks ramped
        rec = {'kernel': ds[0]['name'][:140], 'launches_traced': len(ds), 'avg_launch_us_rocprofv3': round(us, 2),
               'avg_launch_us_rocprofv3_all_calls': round(us_all, 2)}
        if slot in sq and any('SQ_VALU_MFMA_BUSY_CYCLES' in d for d in sq[slot]):
            busy = mean([d['SQ_VALU_MFMA_BUSY_CYCLES'] for d in sq[slot]])
            gui = mean([d['GRBM_GUI_ACTIVE'] for d in sq[slot]])
            flop = busy / 16 * 16384
            rec.update(mfma_busy_cycles=round(busy), mfma_flop_issued_per_launch=round(flop),
                       mfma_util_at_clock=round(busy / (SIMDS * gui / XCDS), 4),
                       mfma_tflops_issued=round(flop / us / 1e6, 1), mfma_frac_of_peak=round(flop / (us * 1e-6) / PEAK_FLOPS, 4),
                       lds_bank_conflict_cycles=round(mean([d.get('SQ_LDS_BANK_CONFLICT', 0) for d in sq[slot]])),
                       wave_cycles_waiting_frac=round(mean([d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES'] for d in sq[slot] if d.get('SQ_WAVE_CYCLES')]), 4))
            if slot in ('gemm_out_proj', 'gemm_c_proj'):  # the alternation really is out_proj (K 768) / c_proj (K 3072)
                spread = [d['SQ_VALU_MFMA_BUSY_CYCLES'] for d in sq[slot]]
                rec['split_consistent'] = bool(max(spread) < 1.5 * min(spread)) if mode_dir.name == 'globals' else None
        if slot in fetch and slot in write:
            rd = 2 * 1024 * mean([d['FETCH_SIZE'] for d in fetch[slot]])
            wr = 1024 * mean([d['WRITE_SIZE'] for d in write[slot]])
            rec.update(hbm_read_bytes_corrected=round(rd), hbm_write_bytes=round(wr), hbm_bytes_per_launch=round(rd + wr),
                       hbm_gbps=round((rd + wr) / us / 1e3, 1), hbm_frac_of_peak=round((rd + wr) / (us * 1e-6) / PEAK_HBM, 4))
        table[slot] = rec
    if mode_dir.name == 'globals':  # operands + output once, 16-bit, at batch 256: T = 12800 rows, C 768, F 3072
        T, C, F = 12800, 768, 3072
        alg = {'gemm_c_fc': 2 * (T * C + F * C + T * F),               # A + W + out
               'gemm_c_proj': 2 * (T * F + C * F + 2 * T * C),          # A + W + x read + x written
               'gemm_out_proj': 2 * (T * C + C * C + 2 * T * C),
               'qkv_attn': 2 * (T * C + 3 * C * C + T * C),             # x + the folded in-projection + the attention output
               'gemm_conv1': 2 * (12544 * 3072 + C * 3072 + 12544 * C),
               'im2col': 256 * 3 * 224 * 224 * 4 + 12544 * 3072 * 2}
        for k_, v_ in alg.items():
            if k_ in table:
                table[k_]['algorithmic_bytes_per_launch'] = v_
                if table[k_].get('hbm_bytes_per_launch'):
                    table[k_]['traffic_over_algorithmic'] = round(table[k_]['hbm_bytes_per_launch'] / v_, 3)
    out[mode_dir.name] = table
    print(mode_dir.name, {k: (v['avg_launch_us_rocprofv3'], v.get('mfma_frac_of_peak'), v.get('hbm_gbps')) for k, v in table.items() if k[0] != '_'})
(ROOT / 'profiles' / f'{tag}_mfma_util_hbm.json').write_text(json.dumps(out, indent=1))
