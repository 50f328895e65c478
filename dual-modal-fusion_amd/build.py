"""Build libdmf_hip.so (the C-ABI library declared in include/dmf.h) for gfx950, in-tree.

    python dual-modal-fusion_amd/build.py [--force] [--stamps] [--shapes FILE]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the tree.
Every translation unit is compiled to its own object (in parallel, only when it or a header changed) and linked.

Extra kernel instances: `--shapes FILE` (or the environment variable DMF_EXTRA_SHAPES) names a text file with one
late-fusion shape per line, `C C2 P S F G` (bands, aux bands, patch, aux scale, gmf.width, spectral groups; H is 64) —
each line adds one row to the v2 kernel's instance table (csrc/dmf_patch_v2.hip: DMF_V2_EXTRA_SHAPES).  A row the
kernel's geometry cannot hold fails at compile time with the static_assert that names the rule.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ('dmf_patch_v2.hip', 'dmf_attention.hip', 'dmf_qua.hip', 'dmf_capi.hip')
SRC = [os.path.join(HERE, 'csrc', n) for n in NAMES]
HDR = [os.path.join(HERE, 'csrc', n) for n in ('dmf_shapes.h', 'dmf_kargs.h', 'dmf_lanes.h', 'dmf_xgmi.h')] + [os.path.join(os.path.dirname(HERE), 'include', 'dmf.h')]
OUT = os.path.join(HERE, 'dmf', 'libdmf_hip.so')
OBJ = os.path.join(HERE, 'build')
# (-Wno-pass-failed: `#pragma unroll` on the run-time class loops of qua_loss_kernel<0> is a request, not a requirement)
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-Wno-pass-failed']
# per-file extras: the reduce launch and the v2 patch kernel take their hot scalars as leading kernel arguments and have
# them preloaded into SGPRs at wave launch (see grad_reduce_kernel, patch_v2_kernel)
EXTRA = {'dmf_capi.hip': ['-mllvm', '-amdgpu-kernarg-preload-count=14'], 'dmf_patch_v2.hip': ['-mllvm', '-amdgpu-kernarg-preload-count=14']}


def extra_shapes(path):
    """`C C2 P S F G` lines -> the X-macro body appended to the v2 instance table."""
    rows = []
    if path:
        for ln in open(path):
            ln = ln.split('#')[0].strip()
            if not ln:
                continue
            v = [int(t) for t in ln.replace(',', ' ').replace('/', ' ').split()]
            if len(v) != 6:
                raise SystemExit('%s: want "C C2 P S F G" per line, got %r' % (path, ln))
            rows.append('X(%d, %d, %d, %d, %d, %d, 64)' % tuple(v))
    return ' '.join(rows)


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=True, stamps=False, shapes=None, defines=(), suffix=None):
    """stamps=True builds the diagnostic variant libdmf_hip_stamps.so (-DDMF_STAMPS; tools/phase_profile_v2.py).
    defines / suffix: an experimental variant libdmf_hip_<suffix>.so with extra -D switches (A/B runs, tools/ab.sh)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    shapes = shapes or os.environ.get('DMF_EXTRA_SHAPES')
    extra = extra_shapes(shapes)
    extra_flags = {k: list(v) for k, v in EXTRA.items()}
    for name, var in (('dmf_patch_v2.hip', 'DMF_V2_FLAGS'), ('dmf_attention.hip', 'DMF_ATTN_FLAGS')):   # A/B variants only
        extra_flags[name] = extra_flags.get(name, []) + os.environ.get(var, '').split()
    defs = (['-DDMF_STAMPS'] if stamps else []) + (['-DDMF_V2_EXTRA_SHAPES(X)=' + extra] if extra else []) + ['-D' + d for d in defines]
    tag = hashlib.sha1(' '.join(FLAGS + defs + [repr(sorted(extra_flags.items()))]).encode()).hexdigest()[:8]
    out = OUT.replace('.so', '_stamps.so') if stamps else OUT
    if suffix:
        out = OUT.replace('.so', '_%s.so' % suffix)
    os.makedirs(OBJ, exist_ok=True)
    objs, jobs = [], []
    for s in SRC:
        o = os.path.join(OBJ, '%s.%s.o' % (os.path.basename(s), tag))
        objs.append(o)
        if force or _newer(o, [s] + HDR + [os.path.abspath(__file__)]):
            jobs.append([hipcc] + FLAGS + extra_flags.get(os.path.basename(s), []) + defs + ['-c', s, '-o', o])
    tagfile = out + '.tag'
    relink = bool(jobs) or _newer(out, objs) or not os.path.exists(tagfile) or open(tagfile).read() != tag
    if not relink:
        return out

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(len(jobs), 5) or 1) as ex:
        list(ex.map(run, jobs))
    run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
    with open(tagfile, 'w') as f:
        f.write(tag)
    return out


if __name__ == '__main__':
    shp = sys.argv[sys.argv.index('--shapes') + 1] if '--shapes' in sys.argv else None
    if '--define' in sys.argv:                   # python build.py --define A=1 [--define B=2] --suffix name
        ds = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == '--define']
        build(shapes=shp, defines=ds, suffix=sys.argv[sys.argv.index('--suffix') + 1])
        sys.exit(0)
    build(force='--force' in sys.argv, shapes=shp)
    if '--stamps' in sys.argv:
        build(stamps=True, shapes=shp)
