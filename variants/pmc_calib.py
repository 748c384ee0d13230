"""Known-traffic kernels to calibrate FETCH_SIZE / WRITE_SIZE: a 512 MiB float copy (reads and
writes 512 MiB each) and a 512 MiB fill (writes only)."""
import torch
x = torch.ones(128 * 1024 * 1024, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    y = x.clone()
    torch.cuda.synchronize()
    y.zero_()
    torch.cuda.synchronize()
