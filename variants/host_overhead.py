"""Host cost of one fused call through the ctypes binding: a batch so small that the GPU is never the
bottleneck, calls issued back to back without synchronisation; wall time per call = what the host spends per call
(Python packing + the library's launches).  Also the Python part alone (the packing functions, no library call)."""
import os, sys, time
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import hip_util as H
from alphasurf_amd import csrc_api as _C, synth

sc = synth.make_sparse_scene(32, 9, seed=3, trunc_a=3.0)
o, d = synth.pinhole_rays(synth.orbit_camera(1), 16, 16)
g, r, op = H.grid_spec(sc), H.rays_spec(o, d), H.opt_spec(synth.SYN_OPTS)
gr = H.zero_grads(g)
gt = torch.rand(o.shape[0], 3, device="cuda")
rgb = torch.zeros_like(gt)
F = synth.SYN_FUSED
args = [dict(F).get(k, dflt) for k, dflt in H.FUSED_ORDER]
def call():
    _C.volume_render_surf_trav_fused(g, r, op, gt, *args, rgb, gr)
for _ in range(50):
    call()
torch.cuda.synchronize()
N = 3000
t0 = time.perf_counter()
for _ in range(N):
    call()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"fused call, 256 rays on 32^3: host {1e6 * (t1 - t0) / N:.1f} us per call issued; {1e6 * (t2 - t0) / N:.1f} us per call completed")
t0 = time.perf_counter()
for _ in range(N):
    _C._pack_grid(g); _C._pack_rays(r); _C._pack_opt(op); _C._pack_grads(gr, g)
t1 = time.perf_counter()
print(f"python packing alone (grid, rays, options, grads): {1e6 * (t1 - t0) / N:.1f} us per call")
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    call()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(graph):
    call()
graph.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    graph.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"the same call replayed from a HIP graph: host {1e6 * (t1 - t0) / N:.1f} us per replay issued; {1e6 * (t2 - t0) / N:.1f} us completed")
