"""Which engine carries a torch device -> pinned copy?  (run under rocprofv3 --kernel-trace --memory-copy-trace)"""
import ctypes
import sys
import time

import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
n = 150 << 20
dev = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
s = torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for k in range(10):
            if mode == "torch":
                host.copy_(dev, non_blocking=True)
            elif mode == "busy":                                  # a kernel right before, like the loop's render
                dev[:4096].fill_(k)
                host.copy_(dev, non_blocking=True)
            else:
                assert hip.hipMemcpyAsync(host.data_ptr(), dev.data_ptr(), n, 2, s.cuda_stream) == 0
    s.synchronize()
    dt = time.perf_counter() - t0
print(mode, "10 x 150 MB in %.2f ms = %.1f GB/s" % (dt * 1e3, 10 * n / dt / 1e9), int(host[0]))
