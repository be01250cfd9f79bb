"""Inter-kernel gaps of the bench from a rocprofv3 --kernel-trace CSV: how much of a step is spent
between kernels.  usage: gap_analysis.py <..._kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
# steady state: the last 60 % of the trace
ev = ev[int(len(ev) * 0.4):]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
small = [g for g in gaps if g < 50_000]
print(f'{len(ev)} kernels over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), '
      f'gaps {sum(gaps) / 1e6:.2f} ms; median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, '
      f'mean of gaps < 50 us {sum(small) / len(small) / 1e3:.2f} us ({len(small)} of {len(gaps)})')
neg = sum(1 for g in gaps if g < 0)
print(f'overlapping launches: {neg}')
