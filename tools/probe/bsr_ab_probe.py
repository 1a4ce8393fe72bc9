"""A/B of library builds on config 4 (ACTIVSg10K BSR-16 x K=128 bf16, the slots kernel) in ONE process on ONE set of operands.
  python tools/probe/bsr_ab_probe.py name=path[:VAR=val;VAR=val...] ...          GPU box only."""
import ctypes
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import datasets, formats, ops, synth  # noqa: E402

VP, U32, I = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int


def main():
    c_bf16 = 0
    if "--c-bf16" in sys.argv:
        sys.argv.remove("--c-bf16")
        c_bf16 = 1
    csr = datasets.load_csr("ACTIVSg10K")
    bsr = formats.csr_to_bsr(csr, 16)
    n = 128
    slots = ops.DeviceBSRCSlots.from_host(bsr)
    b16 = ops.f32_to_bf16(torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda())
    c = torch.empty((csr.num_rows, n), device="cuda", dtype=torch.int16 if c_bf16 else torch.float32)
    stream = torch.cuda.Stream()
    sp = VP(stream.cuda_stream)
    tmp = tempfile.mkdtemp()
    runs, ref = [], None
    for i, spec in enumerate(sys.argv[1:]):
        name, rest = spec.split("=", 1)
        path, _, envs = rest.partition(":")
        env = dict(kv.split("=", 1) for kv in envs.split(";") if kv)
        copy = os.path.join(tmp, f"lib_{i}.so")
        shutil.copy(os.path.join(ROOT, path), copy)
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        lib = ctypes.CDLL(copy)
        lib.mispmm_bsrc_slots_bf16.argtypes = [VP, U32, U32, U32, VP, VP, VP, VP, U32, U32, VP, U32, I]
        lib.mispmm_graph_begin.argtypes = [VP]
        lib.mispmm_graph_end.argtypes = [VP, ctypes.POINTER(VP)]
        lib.mispmm_graph_launch.argtypes = [VP, VP]
        lib.mispmm_last_kernel.restype = ctypes.c_char_p

        def call(lib=lib):
            st = lib.mispmm_bsrc_slots_bf16(sp, csr.num_rows // 16, csr.num_cols, slots.num_steps, VP(slots.extra_ptrs.data_ptr()),
                                            VP(slots.cols.data_ptr()), VP(slots.tiles.data_ptr()), VP(b16.data_ptr()), n, n, VP(c.data_ptr()), n, c_bf16)
            assert st == 0, st
        call()
        torch.cuda.synchronize()
        got = c.clone()
        if ref is None:
            ref = got
        assert torch.equal(got, ref) if c_bf16 else torch.allclose(got, ref, rtol=1e-4, atol=1e-2), name   # variants may add the partial tiles in another (fixed) order
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        assert lib.mispmm_graph_begin(sp) == 0
        for _ in range(1000):
            call()
        g = VP()
        assert lib.mispmm_graph_end(sp, ctypes.byref(g)) == 0
        runs.append((name, lib, g, lib.mispmm_last_kernel().decode()))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _, lib, g, _ in runs:
        for _ in range(5):
            lib.mispmm_graph_launch(g, sp)
    torch.cuda.synchronize()
    times = {name: [] for name, *_ in runs}
    for _ in range(7):
        for name, lib, g, _ in runs:
            with torch.cuda.stream(stream):
                ev0.record(stream)
                for _ in range(4):
                    lib.mispmm_graph_launch(g, sp)
                ev1.record(stream)
            torch.cuda.synchronize()
            times[name].append(ev0.elapsed_time(ev1) * 1e3 / 4000)
    base = np.median(times[runs[0][0]])
    print(f"# ACTIVSg10K BSR-16 x K=128 bf16, C {'bf16' if c_bf16 else 'fp32'}: one process, one set of operands, rounds interleaved")
    for name, _, _, tag in runs:
        t = np.array(times[name])
        print(f"{name:28s} {np.median(t):.3f} us (min {t.min():.3f} max {t.max():.3f})  {100 * (np.median(t) / base - 1):+.1f} %   {tag}")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
