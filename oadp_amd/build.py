"""Build liboake_hip.so (the C-ABI library of include/oake_hip.h) with hipcc for gfx950.

In-tree build: objects go to ``oadp_amd/csrc/_build/``, the shared library to
``oadp_amd/liboake_hip.so`` (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import pathlib
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = pathlib.Path(__file__).resolve().parent
CSRC = ROOT / 'csrc'
BUILD = CSRC / '_build'
LIB = ROOT / 'liboake_hip.so'
SOURCES = ['gemm.hip', 'attention.hip', 'qkv_attn_obj.hip', 'rowops.hip', 'resample.hip', 'jpeg.hip', 'api.hip']
# kernels that lost their A/B: liboake_hip_lab.so only (attn_out: attention + out_proj in one kernel, round 4; qkv_attn: the
# three-images-per-160-row-tile form of the fused qkv + attention kernel, round 5 — the four-image form of qkv_attn_obj.hip won)
LAB_ONLY_SOURCES = ['attn_out.hip', 'qkv_attn.hip']
HEADERS = ['common.h', 'kernels.h', 'attention_head.inc', 'gemm_w8.inc', '../../include/oake_hip.h', '../../include/oake_hip_debug.h']
ARCH = 'gfx950'
# instantiations that may spill: the s_memtime-stamped measurement build of attn_out (oake_debug_attn_out_trace), and
# attn_out itself up to SPILL_SMALL bytes — its out_proj waves sit at the 168-register limit and hipcc parks a few
# epilogue values (one accumulator tile, lane offsets) in scratch: stored once, reloaded once per image, outside the loops
SPILL_OK = ('attn_out_kernelIDF16_Lb1E',)
# qkv_attn_obj_kernel: the DMA waves' tile-end attention task and row group 1's two-tile task park up to 24 registers
# once per TILE, outside the K loop — checked in the ISA (hipcc -S: no scratch_ instruction inside a Depth=2 loop body)
SPILL_SMALL = {'attn_out_kernel': 128, 'qkv_attn_obj_kernel': 128}
FLAGS = [
    f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
    '-Wall', '-Wno-unused-function',
    # MFMA results straight into VGPRs: with the default "AGPR form" hipcc parks accumulators in AGPRs and
    # copies them around wherever a kernel has registers to spare (tools/ubench/mfma_stream.hip: 26 instead
    # of 16.3 cycles per MFMA on a 160-register tile); +0.35 % on the bench, the kernels at the 168-register
    # limit are unaffected
    '-mllvm', '-amdgpu-mfma-vgpr-form=1',
] + os.environ.get('OAKE_EXTRA_FLAGS', '').split()
if os.environ.get('OAKE_LIB_OUT'):  # kernel experiments: build a differently-flagged copy beside the real one
    LIB = pathlib.Path(os.environ['OAKE_LIB_OUT'])
    BUILD = BUILD.parent / ('_build_' + LIB.stem)


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _digest(paths: list[pathlib.Path]) -> str:
    h = hashlib.sha256()
    h.update(' '.join(FLAGS).encode())
    for p in paths:
        h.update(p.read_bytes())
    return h.hexdigest()


LAB_LIB = ROOT / 'liboake_hip_lab.so'
LAB_INCLUDES = ['gemm_lab_q4.inc', 'gemm_lab_duo.inc', 'attention_lab_full.inc']


def _compile(src: str, force: bool, lab: bool = False) -> pathlib.Path:
    s = CSRC / src
    build = BUILD.parent / (BUILD.name + '_lab') if lab else BUILD
    build.mkdir(parents=True, exist_ok=True)
    obj = build / (src + '.o')
    stamp = build / (src + '.sha')
    extra = ['-DOAKE_LAB=1'] if lab else []
    lab_inc = lab or '-DOAKE_LAB=1' in FLAGS
    dig = _digest([s] + [CSRC / hd for hd in HEADERS] + ([CSRC / i for i in LAB_INCLUDES] if lab_inc else [])) + str(lab)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj
    cmd = [_hipcc(), *FLAGS, *extra, '-Rpass-analysis=kernel-resource-usage', '-c', str(s), '-o', str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    # a kernel that spills to scratch puts VMEM round trips into its inner loop (measured: the GEMM
    # loop went from 1600 to 2800 cycles per K-tile with 6 spilled dwords) — refuse to build it
    name = None
    for line in r.stderr.splitlines():
        if 'Function Name:' in line:
            name = line.split('Function Name:')[1].split()[0]
        elif 'ScratchSize [bytes/lane]:' in line:
            n = int(line.split('ScratchSize [bytes/lane]:')[1].split()[0])
            small = max((v for k, v in SPILL_SMALL.items() if k in (name or '')), default=0)
            if n > small and not any(ok in (name or '') for ok in SPILL_OK) and not (
                    os.environ.get('OAKE_LIB_OUT') and os.environ.get('OAKE_ALLOW_SPILL')):  # (experiment builds only)
                raise RuntimeError(f'{src}: kernel {name} spills {n} bytes/lane to scratch')
    other = [l for l in r.stderr.splitlines()
             if 'remark:' not in l and l.strip() and not re.match(r'\s*\d*\s*\|', l)
             and not l.startswith('In file included from')]
    if other:
        sys.stderr.write('\n'.join(other) + '\n')
    stamp.write_text(dig)
    return obj


def _link(objs: list[pathlib.Path], lib: pathlib.Path, force: bool) -> None:
    newest = max(o.stat().st_mtime for o in objs)
    if force or not lib.exists() or lib.stat().st_mtime < newest:
        cmd = [_hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', str(lib), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')


def build_library(force: bool = False, verbose: bool = False, lab: bool = True) -> pathlib.Path:
    """liboake_hip.so — the product: the kernels pick_variant() / the default switches can select — and (lab=True)
    liboake_hip_lab.so beside it: the same sources with -DOAKE_LAB=1, i.e. plus every tile configuration, kernel form
    and measurement epilogue that lost its A/B; loaded only by tools/ and the variant tests (oadp_amd._lib.load_lab).
    An OAKE_LIB_OUT experiment build is a single library with whatever OAKE_EXTRA_FLAGS say."""
    BUILD.mkdir(parents=True, exist_ok=True)
    jobs = [(src, False) for src in SOURCES]
    if lab and not os.environ.get('OAKE_LIB_OUT'):
        jobs += [(src, True) for src in SOURCES if src in ('gemm.hip', 'attention.hip', 'api.hip')]
        jobs += [(src, True) for src in LAB_ONLY_SOURCES]
    if os.environ.get('OAKE_LIB_OUT') and '-DOAKE_LAB=1' in FLAGS:
        # an experiment build of the LAB flavour (OAKE_EXTRA_FLAGS='-DOAKE_LAB=1 ...'): one library, every source with the
        # flag, the lab-only kernels linked in; tools load it through OAKE_LAB_LIB (oadp_amd._lib.load_lab)
        jobs += [(src, False) for src in LAB_ONLY_SOURCES]
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            objs = list(ex.map(lambda j: _compile(j[0], force, j[1]), jobs))
        _link(objs, LIB, force)
        if verbose:
            print(f'built {LIB} (lab flavour)')
        return LIB
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        objs = list(ex.map(lambda j: _compile(j[0], force, j[1]), jobs))
    prod = objs[:len(SOURCES)]
    _link(prod, LIB, force)
    if len(objs) > len(SOURCES):
        labobj = {j[0]: o for j, o in zip(jobs[len(SOURCES):], objs[len(SOURCES):])}
        _link([labobj.get(src, po) for src, po in zip(SOURCES, prod)] + [labobj[src] for src in LAB_ONLY_SOURCES],
              LAB_LIB, force)
    if verbose:
        print(f'built {LIB}' + (f' and {LAB_LIB.name}' if len(objs) > len(SOURCES) else ''))
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv, verbose=True)
