"""Is a hipHostRegister'ed shared-memory segment treated as pinned by torch, and what does a 3.9 MB H2D cost?"""
import ctypes, time
from multiprocessing import shared_memory
import numpy as np, torch
shm = shared_memory.SharedMemory(create=True, size=64 << 20)
arr = np.ndarray((16, 3, 512, 640), np.float32, buffer=shm.buf)
arr[...] = 1.0
dev = torch.empty((16, 3, 512, 640), device="cuda")
torch.cuda.synchronize()
def bench(tag, nb):
    t = [torch.from_numpy(arr[i]) for i in range(16)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(16):
        dev[i].copy_(t[i], non_blocking=nb)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag}: is_pinned={t[0].is_pinned()} enqueue {1e3*(t1-t0)/16:.3f} ms/copy, total {1e3*(t2-t0)/16:.3f} ms/copy = {3.93/((t2-t0)/16)/1e3:.1f} GB/s")
bench("pageable", False)
addr = ctypes.addressof(ctypes.c_char.from_buffer(shm.buf))
rc = torch.cuda.cudart().cudaHostRegister(addr, shm.size, 0)
print("cudaHostRegister rc =", rc, int(rc))
bench("registered, blocking", False)
bench("registered, non_blocking", True)
bench("registered, non_blocking", True)
p = torch.empty((16, 3, 512, 640)).pin_memory()
t0 = time.perf_counter(); torch.cuda.synchronize()
for i in range(16): dev[i].copy_(p[i], non_blocking=True)
torch.cuda.synchronize(); print(f"torch pinned: {3.93/((time.perf_counter()-t0)/16)/1e3:.1f} GB/s")
del arr
try:
    torch.cuda.cudart().cudaHostUnregister(addr)
except Exception as e:
    print("unregister:", e)
shm.close(); shm.unlink()
