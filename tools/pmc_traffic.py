"""Turn rocprofv3 FETCH_SIZE / WRITE_SIZE passes into per-launch HBM traffic per kernel.

MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE (KiB units) counts a wide coalesced read stream at
exactly half its bytes -> doubled here; WRITE_SIZE is uncalibrated (taken as is)."""
import collections, csv, json, sys

def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in fetch:
    if 'oake' not in k:
        continue
    f_kib, n = fetch[k]
    w_kib = write.get(k, (0.0, 0))[0]
    out[k] = {'launches_sampled': n, 'fetch_kib_raw': round(f_kib, 1), 'write_kib_raw': round(w_kib, 1),
              'hbm_read_bytes_corrected': int(2 * f_kib * 1024), 'hbm_write_bytes': int(w_kib * 1024),
              'hbm_bytes_per_launch': int((2 * f_kib + w_kib) * 1024)}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:12]:
    print(f"{v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch  read {v['hbm_read_bytes_corrected']/1e6:8.1f}  write {v['hbm_write_bytes']/1e6:8.1f}  n={v['launches_sampled']:4d}  {k[:110]}")
