"""A/B of library builds / measurement knobs in ONE process on the SAME operands (operand placement moves the headline by up
to 6 % from process to process, profiles/r3/placement_probe.log -- small effects need this).

  python tools/probe/lib_ab_probe.py [--entry uniform|general] [--k-cols 128] name=path[:VAR=val;VAR=val...] ...
Each variant is a copy of the named library loaded under its own name (so its `static const` knobs are read with ITS
environment); every variant multiplies the same A, B into the same C through a 1000-launch graph; rounds are interleaved.
GPU box only."""
import argparse
import ctypes
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import datasets, synth  # noqa: E402

VP, U32, I = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int


def load(path, env, tmp, tag):
    copy = os.path.join(tmp, f"lib_{tag}.so")
    shutil.copy(path, copy)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    lib = ctypes.CDLL(copy)
    lib.mispmm_csr_f32.argtypes = [VP, U32, U32, U32, VP, VP, VP, VP, U32, U32, VP, U32, I, I]
    lib.mispmm_csr_uniform_f32.argtypes = [VP, U32, U32, U32, VP, VP, VP, U32, U32, VP, U32, I]
    lib.mispmm_graph_begin.argtypes = [VP]
    lib.mispmm_graph_end.argtypes = [VP, ctypes.POINTER(VP)]
    lib.mispmm_graph_launch.argtypes = [VP, VP]
    lib.mispmm_last_kernel.restype = ctypes.c_char_p
    return lib, saved


def main():
    p = argparse.ArgumentParser()
    p.add_argument("variants", nargs="+")
    p.add_argument("--entry", default="uniform", choices=["uniform", "general"])
    p.add_argument("--matrix", default="n4c6-b13")
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--acc", type=int, default=0, help="0 reference, 1 fast")
    p.add_argument("--sets", type=int, default=1,
                   help="> 1: every variant's graph rotates over this many distinct (B, C) pairs (0 = 512 MiB of them): the "
                        "HBM-streamed form of the loop (bench.py `hbm_streaming`); graphs hold a whole number of rotations")
    a = p.parse_args()
    csr = datasets.load_csr(a.matrix)
    n = a.k_cols
    dev = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).view(dt)).cuda()  # noqa: E731
    rp, ci, va = dev(csr.row_ptrs.astype(np.uint32), np.int32), dev(csr.col_idxs.astype(np.uint32), np.int32), dev(csr.data.astype(np.float32), np.float32)
    b = torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda()
    c = torch.empty((csr.num_rows, n), device="cuda")
    nsets = a.sets if a.sets > 0 else max(4, -(-(512 << 20) // ((csr.num_cols + csr.num_rows) * n * 4)))
    pairs = [(b, c)] + [(b.clone(), torch.empty_like(c)) for _ in range(nsets - 1)]
    launches = nsets * max(1, -(-1000 // nsets))
    import itertools
    w = int(csr.row_ptrs[1] - csr.row_ptrs[0])
    stream = torch.cuda.Stream()
    sp = VP(stream.cuda_stream)
    tmp = tempfile.mkdtemp()
    runs = []
    for i, spec in enumerate(a.variants):
        name, rest = spec.split("=", 1)
        path, _, envs = rest.partition(":")
        env = dict(kv.split("=", 1) for kv in envs.split(";") if kv)
        lib, saved = load(os.path.join(ROOT, path) if not os.path.isabs(path) else path, env, tmp, f"{i}")

        turn = itertools.count()

        def call(lib=lib, turn=turn):
            b, c = pairs[next(turn) % nsets]
            if a.entry == "uniform":
                st = lib.mispmm_csr_uniform_f32(sp, csr.num_rows, csr.num_cols, w, VP(ci.data_ptr()), VP(va.data_ptr()), VP(b.data_ptr()), n, n,
                                                VP(c.data_ptr()), n, a.acc)
            else:
                st = lib.mispmm_csr_f32(sp, csr.num_rows, csr.num_cols, csr.nnz, VP(rp.data_ptr()), VP(ci.data_ptr()), VP(va.data_ptr()),
                                        VP(b.data_ptr()), n, n, VP(c.data_ptr()), n, 0, a.acc)
            assert st == 0, st
        for _ in range(nsets):                       # the knobs of this copy are read now, with its environment
            call()
        torch.cuda.synchronize()
        tag = lib.mispmm_last_kernel().decode()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        assert lib.mispmm_graph_begin(sp) == 0
        for _ in range(launches):
            call()
        g = VP()
        assert lib.mispmm_graph_end(sp, ctypes.byref(g)) == 0
        runs.append((name, lib, g, tag))
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
                for _ in range(6):
                    lib.mispmm_graph_launch(g, sp)
                ev1.record(stream)
            torch.cuda.synchronize()
            times[name].append(ev0.elapsed_time(ev1) * 1e3 / (6 * launches))
    base = np.median(times[runs[0][0]])
    print(f"# {a.matrix} x K={n} acc={a.acc} entry={a.entry}: one process, "
          + ("one set of operands" if nsets == 1 else f"{nsets} (B, C) pairs in rotation ({nsets * (csr.num_cols + csr.num_rows) * n * 4 / 1e6:.0f} MB: HBM-streamed)")
          + ", rounds interleaved")
    for name, _, _, tag in runs:
        t = np.array(times[name])
        print(f"{name:28s} {np.median(t):.3f} us (min {t.min():.3f} max {t.max():.3f})  {100 * (np.median(t) / base - 1):+.1f} %   {tag}")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
