"""host -> device copy rates of this box: pageable numpy, torch pinned memory, hipHostMalloc flavours"""
import time, ctypes
import numpy as np
import torch
n = 16 * 513 * 513 * 3
a = np.random.default_rng(0).integers(0, 256, n).astype(np.uint8)
dev = torch.empty(n, dtype=torch.uint8, device='cuda')
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
src = torch.as_tensor(a)
dt = t(lambda: dev.copy_(src, non_blocking=True)); print('pageable tensor -> device        %.2f ms  %.1f GB/s' % (dt * 1e3, n / dt / 1e9))
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True); pin.copy_(src)
dt = t(lambda: dev.copy_(pin, non_blocking=True)); print('torch pin_memory -> device       %.2f ms  %.1f GB/s' % (dt * 1e3, n / dt / 1e9))
pin2 = src.clone().pin_memory()
dt = t(lambda: dev.copy_(pin2, non_blocking=True)); print('tensor.pin_memory() -> device    %.2f ms  %.1f GB/s' % (dt * 1e3, n / dt / 1e9))
dt = t(lambda: pin.copy_(src)); print('pageable -> pinned (CPU)         %.2f ms  %.1f GB/s' % (dt * 1e3, n / dt / 1e9))
hip = ctypes.CDLL('libamdhip64.so')
for flags, name in ((0, 'hipHostMallocDefault'), (0x2, 'hipHostMallocMapped'), (0x40000000, 'hipHostMallocNonCoherent'), (0x80000000, 'hipHostMallocCoherent')):
    p = ctypes.c_void_p()
    rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(n), ctypes.c_uint(flags))
    if rc != 0:
        print(name, 'rc', rc); continue
    ctypes.memmove(p, a.ctypes.data, n)
    st = torch.cuda.current_stream().cuda_stream
    def cp():
        hip.hipMemcpyAsync(ctypes.c_void_p(dev.data_ptr()), p, ctypes.c_size_t(n), ctypes.c_int(1), ctypes.c_void_p(st))
    dt = t(cp); print('%-32s %.2f ms  %.1f GB/s' % (name + ' -> device', dt * 1e3, n / dt / 1e9))
    t0 = time.perf_counter()
    for _ in range(5):
        ctypes.memmove(p, a.ctypes.data, n)
    print('    CPU memmove into it          %.2f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
    hip.hipHostFree(p)
