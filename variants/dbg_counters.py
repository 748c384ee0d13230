"""Pipeline counters of one fused call on the bench scene (ASURF_DEBUG=16 prints them)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["ASURF_DEBUG"] = "16"
import hip_util
from alphasurf_amd import lib as _lib, synth
width = int(sys.argv[1]) if len(sys.argv) > 1 else 800
sc = synth.make_sparse_scene(256, 9, trunc_a=5.0, **synth.BENCH_SCENE)
o, d = synth.pinhole_rays(synth.orbit_camera(0), width, width)
gt = np.random.RandomState(7).rand(o.shape[0], 3).astype(np.float32)
fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4, lambda_conv_mode_samp=1e-6)
for mode in ("sweep", "step"):
    _lib.config_set("ASURF_WALK", mode)
    print(mode, flush=True)
    hip_util.render_fused(sc, o, d, synth.SYN_OPTS, gt, fused)
