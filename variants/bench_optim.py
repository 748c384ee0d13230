"""Times the N1 optimizer steps at the bench scene's sizes (1.31 M rows): dense, row-mask (20 % touched) and
row-index (5 % touched) variants.  Model: 12 B read + 12 B written per updated element."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphasurf_amd import csrc_api as _C

dev = torch.device("cuda:0")
n = 1309130
out = {}
for cols, name in ((27, "sh_data [n,27]"), (48, "sh_data [n,48]"), (1, "density_data [n,1]")):
    data = torch.randn(n, cols, device=dev)
    rms = torch.rand(n, cols, device=dev)
    grad = torch.randn(n, cols, device=dev)
    mask = torch.rand(n, device=dev) < 0.2
    idx = torch.nonzero(torch.rand(n, device=dev) < 0.05).view(-1)
    for kind, indexer, rows in (("dense", torch.empty((), device=dev), n), ("mask 20%", mask, int(mask.sum())),
                                ("index 5%", idx, idx.numel())):
        def fn():
            _C.rmsprop_step(data, rms, grad, indexer, 0.95, 1e-3, 1e-8, -1e9, 1e-3)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 50
        out[f"rmsprop {name} {kind}"] = dict(ms=round(ms, 4), updated_elements=rows * cols,
                                             model_GBps=round(rows * cols * 24 / ms / 1e6, 1))
print(json.dumps(out, indent=1))
