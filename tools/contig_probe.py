"""Does PHYSICALLY CONTIGUOUS device memory (hipExtMallocWithFlags(hipDeviceMallocContiguous)) change what the overlay gets?
Times cama_overlay_probe (stamp-free overlay, 40 frames of 6 x 1600x900 -> 2x3 mosaics: 1 GB read + 1 GB written) over K pairs of
buffers allocated the plain way (hipMalloc) and the contiguous way: each pair alone (warm: the same pair again and again), and
cycling through the pairs (cold: every launch walks buffers that were last touched K launches ago).
Usage: python tools/contig_probe.py [K=6]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cama_amd import _lib

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
F, C, H, W, cols = 40, 6, 900, 1600, 3
nbytes = F * C * H * W * 3
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
torch.cuda.init(); torch.zeros(1, device="cuda:0")
L = _lib.lib()
stream = torch.cuda.current_stream().cuda_stream


def alloc(contig):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, 0x4) if contig else hip.hipMalloc(ctypes.byref(p), nbytes)
    if rc != 0 or not p.value:
        raise RuntimeError(f"allocation failed rc={rc} contiguous={contig}")
    hip.hipMemset(p, 1, nbytes)
    return p.value


def probe(src, dst, reps):
    ms = ctypes.c_double()
    _lib.check(L.cama_overlay_probe(src, dst, F, C, H, W, cols, reps, ctypes.byref(ms), stream))
    return ms.value


for contig in ([False] if os.environ.get("PROBE_PLAIN_ONLY") else [False, True, False, True]):
    try:
        pairs = [(alloc(contig), alloc(contig)) for _ in range(K)]
    except RuntimeError as e:
        print(e); continue
    torch.cuda.synchronize()
    warm = [probe(s, d, 6) for s, d in pairs]
    cold = []
    for rnd in range(6):
        for s, d in pairs:
            cold.append(probe(s, d, 1))
    cold = np.array(cold[K:]).reshape(-1, K)               # (drop the first round)
    frac = lambda ms: 2 * nbytes / (ms * 1e-3) / 8e12
    print(f"{'contiguous' if contig else 'plain     '}: warm per pair " + " ".join(f"{frac(m):.3f}" for m in warm) +
          f" | cycling through the {K} pairs, mean per pair " + " ".join(f"{frac(m):.3f}" for m in cold.mean(axis=0)) +
          f" | warm mean {frac(np.mean(warm)):.3f} cold mean {frac(cold.mean()):.3f}")
    for s, d in pairs:
        hip.hipFree(s); hip.hipFree(d)
