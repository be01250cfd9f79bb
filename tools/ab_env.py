"""A/B of bench configurations in ONE GPU session (boxes differ by +-3 %): each configuration is a label plus
environment settings (OAKE_LIB=<other build>, OAKE_GEMM_VARIANT=.., OAKE_PATCH_DIRECT=0, ...); rounds are
interleaved, medians printed.
usage: ab_env.py <rounds> label[:VAR=val[,VAR=val...]] ...        e.g.  ab_env.py 5 base:OAKE_LIB=oadp_amd/liboake_base.so new"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1])
configs = []
for spec in sys.argv[2:]:
    label, _, envs = spec.partition(':')
    configs.append((label, dict(kv.split('=', 1) for kv in envs.split(',') if kv)))
res = {label: [] for label, _ in configs}
one = {label: [] for label, _ in configs}
extra = os.environ.get('AB_BENCH_ARGS', '--no-cpu-baseline --no-modes --steps 40').split()
for r in range(rounds):
    for label, env in configs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *extra], env=dict(os.environ, OAKE_BENCH_FULL_LINE='1', **env),
                             capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
        if not line:
            print(label, 'FAILED', out.stderr[-800:], flush=True)
            continue
        d = json.loads(line[-1])
        res[label].append(d['value'])
        if d.get('one_lane_images_per_sec'):
            one[label].append(d['one_lane_images_per_sec'])
        print(f'round {r} {label}: {d["value"]}  one-lane {d.get("one_lane_images_per_sec")}', flush=True)
for label, _ in configs:
    if res[label]:
        print(f'{label:14s} median {statistics.median(res[label]):10.1f}  (min {min(res[label]):.1f} max {max(res[label]):.1f})'
              + (f'  one-lane median {statistics.median(one[label]):.1f}' if one[label] else ''), flush=True)
