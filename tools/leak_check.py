"""Create and destroy many handles of different shapes and watch the free device memory (leak check)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from seigen_amd import _lib
from seigen_amd.backend import HipBlock
def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
f0 = free()
for it in range(150):
    dim = 2 + it % 2
    n = (24, 20) if dim == 2 else (16, 6, 5)
    mask = 0 if it % 3 else (0b1000 if dim == 2 else 0b110000)
    blk = HipBlock(dim, 1 + it % 4, n, (0.1,) * dim, (0.0,) * dim, "left", mask)
    blk.set_params(1.0, 1e-4, 0.5, 0.25)
    if mask == 0:
        blk.set_absorption(np.ones((blk.ncells, 15 if dim == 2 else 35)), 4)
        blk.set_source([3, 7], np.ones((4, 2, dim, dim)))
        blk.step(4)
    blk.close()
    if it % 50 == 49:
        print(it + 1, "handles: free memory change %.1f MiB" % (free() - f0))
