"""bench.py — OAKE images/sec for ViT-B/32 encode_image at 224^2, batch 256 per GPU.

A "step" is one pass of the hot path (oadp.oake.globals' encode_image + fused L2-normalise + fp16
cast) over one batch of 256 synthetic device-resident 3x224x224 crops, random-init ViT-B/32 weights.
Prints ONE JSON line (rank 0).  Multi-GPU: one process per GPU, images sharded, no data-path
collective; RCCL only for the barrier, the max-over-ranks time and the counters gather.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

FLOP_PER_IMAGE = 8_817_623_040          # BASELINE.md §3 (2 FLOP/MAC; LN/softmax/GELU/bias excluded)
# The library runs the last block's query / attention output / out_proj / MLP for the CLS row only
# (the only row ln_post reads; identical embeddings): 49 of 50 rows of those GEMMs and of the attention
# are not executed.  Roofline fractions are quoted on the FLOPs that ARE executed.
FLOP_SKIPPED_LAST_BLOCK = 49 * (2 * 2 * 768 * 768 + 2 * 2 * 768 * 3072) + 4 * 49 * 50 * 64 * 12
FLOP_PER_IMAGE_EXECUTED = FLOP_PER_IMAGE - FLOP_SKIPPED_LAST_BLOCK
PEAK_MFMA_DENSE = 2.5e15                # MI355X bf16/f16 dense (MI355X_MICROARCH.md)


def cpu_baseline(sd, seconds: float = 12.0) -> dict:
    """The oracle's fp32 torch-CPU encode_image (a PORT/restatement: the reference's `clip` fork is
    not importable anywhere — SURVEY.md §8c) on a bounded sample of the same workload."""
    from oadp_amd.weights import synthetic_images
    from oracle.vit_ref import ViTConfig, encode_image_ref, l2_normalize
    bs = 32
    x = synthetic_images(bs, seed=5)
    cfg = ViTConfig()
    threads = torch.get_num_threads()
    l2_normalize(encode_image_ref(sd, cfg, x[:4]))  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        l2_normalize(encode_image_ref(sd, cfg, x)).half()
        n += bs
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 256:
            break
    return {'value': round(n / dt, 2), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{n} synthetic 3x224x224 images in batches of {bs}, fp32 torch-CPU oracle, '
                      f'{dt:.1f} s'}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--dtype', choices=['f16', 'bf16'], default=os.environ.get('OAKE_DTYPE', 'f16'))
    ap.add_argument('--residual', choices=['f16', 'f32'], default='f16',
                    help='residual-stream element type (f16 = compute dtype, as the reference GPU model)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    dist = world > 1
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.device('cuda', local % torch.cuda.device_count())
    if dist:
        import torch.distributed as td
        # RCCL on ROCm; OAKE_BENCH_BACKEND=gloo lets two ranks share one GPU to exercise this path on a 1-GPU box
        td.init_process_group(backend=os.environ.get('OAKE_BENCH_BACKEND', 'nccl'))

    from oadp_amd import _lib, clip
    from oadp_amd.weights import synthetic_state_dict
    if 'OAKE_GEMM_VARIANT' in os.environ:  # A/B runs of the GEMM tile configurations
        _lib.load().oake_debug_set_gemm_variant(int(os.environ['OAKE_GEMM_VARIANT']))
    if 'OAKE_CLS_LAST' in os.environ:  # 0: run the last block for every token, as the reference does
        _lib.load().oake_debug_set_cls_last(int(os.environ['OAKE_CLS_LAST']))
    if 'OAKE_ATTN_VARIANT' in os.environ:
        _lib.load().oake_debug_set_attention_variant(int(os.environ['OAKE_ATTN_VARIANT']))
    cdt = torch.float16 if args.dtype == 'f16' else torch.bfloat16
    sd = synthetic_state_dict()
    model, _ = clip.load(sd, compute_dtype=cdt, max_batch=args.batch,
                         residual_dtype=torch.float32 if args.residual == 'f32' else None)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(args.batch, 3, 224, 224, generator=g, device=dev)  # resident in HBM

    # consecutive steps alternate over two lanes = (native handle, HIP stream) pairs: a step is still one
    # pass over one batch of 256, but the kernels of step k+1 fill the start-up / tail bubbles of step k
    # (OAKE_BENCH_LANES=1: one stream, every kernel of a step strictly after the previous step's)
    n_lanes = max(1, int(os.environ.get('OAKE_BENCH_LANES', 2)))
    lane_streams = [torch.cuda.Stream(dev) for _ in range(n_lanes)]
    step_no = [0]

    def step():
        lane = step_no[0] % n_lanes
        step_no[0] += 1
        model.visual.lane = lane
        try:
            with torch.cuda.stream(lane_streams[lane]):
                return model.encode_image(images, normalize=True, out_dtype=torch.float16)
        finally:
            model.visual.lane = 0

    for _ in range(n_lanes):  # set-up, not a step of the contract: create every lane's handle (weights, buffers)
        out = step()
    torch.cuda.synchronize()
    step_no[0] = 0
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(out.float()).all()

    counters = torch.tensor([args.batch * args.steps, args.batch * args.steps, elapsed,
                             out.numel() * 2 * args.steps], dtype=torch.float64, device=dev)
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = tmax.item()
        gathered = [torch.zeros_like(counters) for _ in range(world)]
        td.all_gather(gathered, counters)  # the one RCCL exchange: 32 B per rank
        total_images = sum(c[0].item() for c in gathered)
    else:
        total_images = counters[0].item()

    roofline = None
    kernels = None
    if rank == 0 and not args.no_profile:
        torch.cuda.synchronize()
        model.visual.profile(True)  # (lane 0's handle, on the current stream: kernels one after another)
        for _ in range(3):
            model.encode_image(images, normalize=True, out_dtype=torch.float16)
        prof = model.visual.profile_read()
        model.visual.profile(False)
        gemms = [p for p in prof if p['flops'] > 0 and p['name'].startswith('gemm')]
        dom = max(gemms, key=lambda p: p['total_ms'])
        achieved = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
        roofline = {
            'bound': 'mfma', 'kernel': dom['name'],
            'achieved': round(achieved, 1), 'peak': PEAK_MFMA_DENSE / 1e12, 'unit': 'TFLOP/s',
            'frac': round(achieved * 1e12 / PEAK_MFMA_DENSE, 4),
            'avg_launch_us': round(dom['total_ms'] * 1e3 / dom['launches'], 2),
            'traffic': None,
        }
        # HBM bytes per launch come from separate rocprofv3 --pmc passes over this same command
        # (tools/pmc_traffic.py -> profiles/hbm_traffic.json); PMC cannot be sampled from in here.
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath) and args.batch == 256 and args.dtype == 'f16':
            rec = json.load(open(tpath)).get(dom['name'])
            if rec:
                roofline['traffic'] = rec['hbm_bytes_per_launch']
                roofline['traffic_unit'] = 'bytes/launch (PMC, profiles/hbm_traffic.json)'
        # the committed rocprofv3 --kernel-trace --stats summary of this same command, for comparison
        # (kernel begin/end stamps of consecutive launches include the hand-over between kernels, which
        # rocprofv3's per-dispatch interval does not: expect the live number a few % higher)
        spath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_rocprofv3_kernel_stats.csv')
        rec = (json.load(open(tpath)).get(dom['name']) or {}) if os.path.exists(tpath) else {}
        if os.path.exists(spath) and rec.get('kernel') and args.batch == 256 and args.dtype == 'f16':
            import csv
            for row in csv.DictReader(open(spath)):
                if row['Name'] == rec['kernel']:
                    roofline['avg_launch_us_rocprofv3'] = round(float(row['AverageNs']) / 1e3, 2)
                    roofline['rocprofv3_summary'] = 'profiles/r01_rocprofv3_kernel_stats.csv'
        tot_ms = sum(p['total_ms'] for p in prof)
        kernels = {p['name']: {'ms_per_step': round(p['total_ms'] / 3, 4),
                               'share': round(p['total_ms'] / tot_ms, 4),
                               'tflops': round(p['flops'] / (p['total_ms'] * 1e-3) / 1e12, 1) if p['flops'] else None}
                   for p in sorted(prof, key=lambda p: -p['total_ms'])}

    if rank == 0:
        value = total_images / elapsed
        flop_image = FLOP_PER_IMAGE if os.environ.get('OAKE_CLS_LAST') == '0' else FLOP_PER_IMAGE_EXECUTED
        line = {
            'metric': 'OAKE images/sec (ViT-B/32, 224^2, bs256)', 'value': round(value, 1),
            'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'oadp.oake.globals: ViT-B/32 encode_image + L2-normalise + fp16, '
                                   f'single 224^2 crop per image, batch {args.batch} per GPU, '
                                   'random-init weights, device-resident N(0,1) inputs'
                                   + ('' if os.environ.get('OAKE_CLS_LAST') == '0' else
                                      '; last block evaluated for the CLS rows only (the rows ln_post reads: '
                                      'identical embeddings, 6.6 % fewer FLOPs; OAKE_CLS_LAST=0 runs every row)'),
                       'batch_per_gpu': args.batch, 'sharding': f'images x{world} (no data-path collective)',
                       'hip_streams': n_lanes},
            'mfma_roofline_frac_e2e': round(value / world * flop_image / PEAK_MFMA_DENSE, 4),
            'flop_per_image': {'model': FLOP_PER_IMAGE, 'executed': flop_image},
            'roofline': roofline,
            'kernels': kernels,
            # (rank 0 at N=1 only: with more ranks the other processes would sit in teardown for its 10+ s)
            'cpu_baseline': None if (args.no_cpu_baseline or world > 1) else cpu_baseline(sd),
        }
        print(json.dumps(line), flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == '__main__':
    main()
