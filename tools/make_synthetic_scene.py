"""Write the synthetic scene of BASELINE.json configs[1] (or another size) as ms4.tif.npy / pan.tif.npy / label.npy.

    python tools/make_synthetic_scene.py <dir> [H W C C2 S n_classes seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dual-modal-fusion_amd'))
from dmf import synth  # noqa: E402

if __name__ == '__main__':
    d = sys.argv[1]
    H, W, C, C2, S, ncls, seed = ([int(v) for v in sys.argv[2:9]] + [145, 145, 200, 1, 1, 16, 0][len(sys.argv) - 2:])[:7]
    synth.write_scene(d, *synth.make_scene(H, W, C, C2, S, n_classes=ncls, seed=seed))
    print('wrote', d)
