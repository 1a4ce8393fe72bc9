"""Does the kind of memory B (and C) live in change the headline?  hipMalloc (coarse-grained: an XCD's L2 copy is only valid
until the next kernel boundary) against hipExtMallocWithFlags(hipDeviceMallocFinegrained / Uncached).  One process, same
kernel, 1000-launch graphs.   python tools/probe/finegrained_probe.py     GPU box only."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, synth  # noqa: E402

VP = ctypes.c_void_p
FLAGS = {"default (hipMalloc)": None, "finegrained": 0x1, "uncached": 0x3}


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(VP), ctypes.c_size_t, ctypes.c_uint]
    hip.hipMalloc.argtypes = [ctypes.POINTER(VP), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [VP, VP, ctypes.c_size_t, ctypes.c_int]
    l = capi.lib()
    csr = datasets.load_csr("n4c6-b13")
    n = 128
    w = int(csr.row_ptrs[1])
    ci = torch.from_numpy(csr.col_idxs.astype(np.uint32).view(np.int32)).cuda()
    va = torch.from_numpy(csr.data.astype(np.float32)).cuda()
    bh = np.ascontiguousarray(synth.dense_b(csr.num_cols, n))
    stream = torch.cuda.Stream()
    sp = VP(stream.cuda_stream)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def alloc(nbytes, flag):
        p = VP()
        rc = hip.hipMalloc(ctypes.byref(p), nbytes) if flag is None else hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, flag)
        assert rc == 0, rc
        return p

    ref = None
    print("# n4c6-b13 x K=128 REFERENCE, uniform-row entry point, 1000-launch graphs, us per SpMM")
    for bname, bflag in FLAGS.items():
        for cname, cflag in (("default (hipMalloc)", None), ("finegrained", 0x1)):
            bp, cp = alloc(bh.nbytes, bflag), alloc(csr.num_rows * n * 4, cflag)
            assert hip.hipMemcpy(bp, VP(bh.ctypes.data), bh.nbytes, 1) == 0

            def call():
                capi.check(l.mispmm_csr_uniform_f32(sp, csr.num_rows, csr.num_cols, w, VP(ci.data_ptr()), VP(va.data_ptr()), bp, n, n, cp, n, 0))
            call()
            torch.cuda.synchronize()
            got = np.empty((csr.num_rows, n), np.float32)
            assert hip.hipMemcpy(VP(got.ctypes.data), cp, got.nbytes, 2) == 0
            if ref is None:
                ref = got
            assert np.array_equal(got, ref)
            capi.check(l.mispmm_graph_begin(sp))
            for _ in range(1000):
                call()
            g = VP()
            capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
            for _ in range(5):
                capi.check(l.mispmm_graph_launch(g, sp))
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                with torch.cuda.stream(stream):
                    ev0.record(stream)
                    for _ in range(6):
                        capi.check(l.mispmm_graph_launch(g, sp))
                    ev1.record(stream)
                torch.cuda.synchronize()
                ts.append(ev0.elapsed_time(ev1) * 1e3 / 6000)
            print(f"B {bname:20s} C {cname:20s} {np.median(ts):.3f} us (min {min(ts):.3f} max {max(ts):.3f})")


if __name__ == "__main__":
    main()
