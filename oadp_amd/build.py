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
SOURCES = ['gemm.hip', 'attention.hip', 'attn_out.hip', 'rowops.hip', 'resample.hip', 'jpeg.hip', 'api.hip']
HEADERS = ['common.h', 'kernels.h', '../../include/oake_hip.h', '../../include/oake_hip_debug.h']
ARCH = 'gfx950'
# instantiations that may spill: the s_memtime-stamped measurement build of attn_out (oake_debug_attn_out_trace), and
# attn_out itself up to SPILL_SMALL bytes — its out_proj waves sit at the 168-register limit and hipcc parks a few
# epilogue values (one accumulator tile, lane offsets) in scratch: stored once, reloaded once per image, outside the loops
SPILL_OK = ('attn_out_kernelIDF16_Lb1E',)
SPILL_SMALL = {'attn_out_kernel': 128}
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


def _compile(src: str, force: bool) -> pathlib.Path:
    s = CSRC / src
    obj = BUILD / (src + '.o')
    stamp = BUILD / (src + '.sha')
    dig = _digest([s] + [CSRC / hd for hd in HEADERS])
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj
    cmd = [_hipcc(), *FLAGS, '-Rpass-analysis=kernel-resource-usage', '-c', str(s), '-o', str(obj)]
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
            if n > small and not any(ok in (name or '') for ok in SPILL_OK):
                raise RuntimeError(f'{src}: kernel {name} spills {n} bytes/lane to scratch')
    other = [l for l in r.stderr.splitlines()
             if 'remark:' not in l and l.strip() and not re.match(r'\s*\d*\s*\|', l)]
    if other:
        sys.stderr.write('\n'.join(other) + '\n')
    stamp.write_text(dig)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> pathlib.Path:
    BUILD.mkdir(parents=True, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    newest = max(o.stat().st_mtime for o in objs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest:
        cmd = [_hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'built {LIB}')
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv, verbose=True)
