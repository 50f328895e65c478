"""Build libdmf_hip.so (the C-ABI library declared in include/dmf.h) for gfx950, in-tree.

    python dual-modal-fusion_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, 'csrc', n) for n in ('dmf_patch_kernel.hip', 'dmf_patch_v2.hip', 'dmf_attention.hip', 'dmf_qua.hip', 'dmf_capi.hip')]
HDR = [os.path.join(HERE, 'csrc', n) for n in ('dmf_shapes.h', 'dmf_kargs.h', 'dmf_lanes.h', 'dmf_xgmi.h')] + [ os.path.join(os.path.dirname(HERE), 'include', 'dmf.h')]
OUT = os.path.join(HERE, 'dmf', 'libdmf_hip.so')
# (-Wno-pass-failed: `#pragma unroll` on the run-time class loops of qua_loss_kernel<0> is a request, not a requirement)
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unused-function', '-Wno-pass-failed']


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(p) <= t for p in SRC + HDR + [os.path.abspath(__file__)])


def build(force=False, verbose=True, stamps=False):
    """stamps=True builds the diagnostic variant libdmf_hip_stamps.so (-DDMF_STAMPS; tools/phase_profile.py)."""
    if not stamps and not force and up_to_date():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    out = OUT.replace('.so', '_stamps.so') if stamps else OUT
    cmd = [hipcc] + FLAGS + (['-DDMF_STAMPS'] if stamps else []) + SRC + ['-o', out]
    if stamps:
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return out
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    if '--stamps' in sys.argv:
        build(stamps=True)
