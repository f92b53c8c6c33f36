"""Minimal record for a claim of round 3 (DESIGN/LAB_NOTES: "memset nodes inside a replayed HIP graph did not reliably re-zero
their buffer on this ROCm build"; the library zero-fills with a kernel since).  A graph holding nothing but
    memset(buf, 0)  ->  buf += 1 (kernel)  ->  out = buf (copy)
is replayed many times; every replay must read back exactly 1 everywhere.  Several buffer sizes, the memset issued through
torch (`zero_()`, which becomes hipMemsetAsync under capture) and through hipMemsetAsync called from C via ctypes on the capturing
stream (how the library used to issue it).  Prints one line per case; no kernel of this repository is involved."""
import ctypes
import sys

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def case(nbytes, how, replays=300, interleave=False):
    d = torch.device("cuda", 0)
    n = nbytes // 4
    buf = torch.full((n,), 7.0, device=d)
    out = torch.empty_like(buf)
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=d)   # 64 MB streamed between replays: evicts the L2s
    side = torch.cuda.Stream(d)
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            if how == "torch":
                buf.zero_()
            else:
                rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(nbytes),
                                        ctypes.c_void_p(torch.cuda.current_stream(d).cuda_stream))
                assert rc == 0, rc
            buf.add_(1.0)
            out.copy_(buf)
        bad = 0
        worst = 0.0
        for i in range(replays):
            if interleave:
                noise.fill_(i & 255)
            graph.replay()
            if i % 10 == 9 or i < 5:
                side.synchronize()
                m = float((out - 1.0).abs().max())
                if m != 0.0:
                    bad += 1
                    worst = max(worst, m)
        side.synchronize()
    print("memset via %-5s %9d bytes, %d replays%s: %s" % (how, nbytes, replays, " with L2 churn" if interleave else "",
                                                            "always re-zeroed" if bad == 0 else "%d checks read a stale buffer (max |out-1| = %g)" % (bad, worst)))
    return bad


if __name__ == "__main__":
    total = 0
    for nbytes in (4096, 1 << 20, 1536000, 33554432):
        for how in ("torch", "hip"):
            total += case(nbytes, how)
    total += case(1536000, "hip", interleave=True)
    sys.exit(0)
