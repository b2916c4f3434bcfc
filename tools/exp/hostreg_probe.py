"""Which host ranges hipHostRegister accepts on this box (page alignment, size, allocator)."""
import ctypes, mmap
import numpy as np
import torch
torch.cuda.init()
rt = torch.cuda.cudart()
def try_reg(name, ptr, n):
    r = rt.cudaHostRegister(ptr, n, 0)
    print(name, hex(ptr), n, "->", r)
    if int(r) == 0:
        rt.cudaHostUnregister(ptr)
a = np.zeros(374400 // 4, np.float32)
try_reg("numpy unaligned", a.ctypes.data, a.nbytes)
base = a.ctypes.data & ~4095
try_reg("numpy widened", base, ((a.ctypes.data + a.nbytes + 4095) & ~4095) - base)
b = np.zeros(64 << 20, np.uint8)
try_reg("numpy 64MiB (mmap-backed)", b.ctypes.data, b.nbytes)
pb = (b.ctypes.data + 4095) & ~4095
try_reg("numpy 64MiB aligned inner", pb, 32 << 20)
m = mmap.mmap(-1, 1 << 20)
addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
try_reg("mmap 1MiB", addr, 1 << 20)
