"""Multi-GPU path on CPU: world_size 2 and 3 over the gloo backend.  Exercises the row partition,
the padded slab ring, bucketed all-gather and reassembly of mispmm.dist.ShardedCsrSpmm with the
oracle injected as the per-rank compute step (tests only -- the product default is the HIP kernel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, matrix, n, bucket, steps, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "cuda-optimization-for-spmm_amd")]
    from mispmm import datasets, synth
    from mispmm import dist as mdist
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        csr = datasets.load_csr(matrix)
        calls = []

        def compute(a, b, out):     # the oracle stands in for the HIP kernel on CPU
            rp = a.row_ptrs.numpy().view(np.uint32)
            ci = a.col_idxs.numpy().view(np.uint32)
            out.copy_(torch.from_numpy(orc.spmm_csr(rp, ci, a.data.numpy(), b.numpy())))
            calls.append(1)

        job = mdist.ShardedCsrSpmm(csr, n, device="cpu", bucket=bucket, compute=compute)
        b_host = synth.dense_b(csr.num_cols, n) if rank == 0 else None
        job.broadcast_b(b_host)                                   # only rank 0 has B before this
        full_b = synth.dense_b(csr.num_cols, n)
        assert np.array_equal(job.b.numpy(), full_b)
        job.run(steps)
        job.finish()
        assert len(calls) == steps
        ref = orc.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, full_b)
        got = job.gathered_c().numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref), "gathered C differs from the unsharded product"
        # slab bookkeeping: contiguous, covering, nnz-balanced
        b = job.bounds
        assert b[0] == 0 and b[-1] == csr.num_rows and np.all(np.diff(b) >= 0)
        nnz = np.diff(csr.row_ptrs.astype(np.int64))
        per = [int(nnz[b[r]:b[r + 1]].sum()) for r in range(world)]
        assert max(per) - min(per) <= max(int(nnz.max()), 1) * 2
        # a second run continues cleanly after a partial bucket, and no-gather steps leave the gather alone
        job.run(bucket + 1)
        job.finish()
        assert np.array_equal(job.gathered_c().numpy(), ref)
        job.run(2, gather=False)
        job.finish(gather=False)
        assert np.array_equal(job.local_slab().numpy(), ref[b[rank]:b[rank + 1]])
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,matrix,n,bucket,steps", [(2, "n3c5-b6", 8, 4, 6), (2, "qh1484", 16, 3, 3),
                                                         (3, "Hamrle1", 5, 2, 5), (2, "n4c6-b13", 32, 2, 3)])
def test_sharded_spmm_gloo(tmp_path, world, matrix, n, bucket, steps):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, matrix, n, bucket, steps, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def _ell_worker(rank, world, port, matrix, n, bucket, steps, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "cuda-optimization-for-spmm_amd")]
    from mispmm import datasets, formats, synth
    from mispmm import dist as mdist
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        csr = datasets.load_csr(matrix)
        ellc = formats.csr_to_ell_colmajor(csr)
        rows = np.repeat(np.arange(csr.num_rows, dtype=np.uint32), np.diff(csr.row_ptrs.astype(np.int64)))

        def compute(a, b, out):     # the oracle's ELL arithmetic (fp32 product, fp32 add in slot order) on the rank's rows
            cols = np.asarray(a.col_idxs, dtype=np.uint32).reshape(a.num_rows, -1)
            vals = np.asarray(a.data, dtype=np.float32).reshape(a.num_rows, -1)
            live = cols != 0xFFFFFFFF
            r = np.repeat(np.arange(a.num_rows, dtype=np.uint32), live.sum(axis=1))
            out.copy_(torch.from_numpy(orc.spmm_coo(a.num_rows, r, cols[live], vals[live], b.numpy())))

        job = mdist.ShardedEllSpmm(ellc, n, device="cpu", bucket=bucket, compute=compute)
        job.broadcast_b(synth.dense_b(csr.num_cols, n) if rank == 0 else None)
        job.run(steps)
        job.finish()
        full_b = synth.dense_b(csr.num_cols, n)
        ref = orc.spmm_ell_colmajor(ellc.num_rows, ellc.row_idxs, ellc.data, full_b)
        assert np.array_equal(job.gathered_c().numpy(), ref), "gathered C differs from the unsharded ELL product"
        assert np.array_equal(ref, orc.spmm_coo(csr.num_rows, rows, csr.col_idxs, csr.data, full_b))
        b = job.bounds
        assert b[0] == 0 and b[-1] == csr.num_rows and np.all(np.diff(b) >= 0) and job.local_nnz > 0
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,matrix,n,bucket,steps", [(2, "qh1484", 16, 3, 4), (3, "n3c5-b6", 8, 2, 3)])
def test_sharded_ell_gloo(tmp_path, world, matrix, n, bucket, steps):
    """ShardedEllSpmm (ELL by rows) over gloo with the oracle as the per-rank compute step: partition
    by occupied slots, padded slab ring, bucketed all-gather, reassembly -- the gathered C equals the unsharded ELL product."""
    port = _free_port()
    mp.spawn(_ell_worker, args=(world, port, matrix, n, bucket, steps, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_row_slice_is_a_standalone_csr():
    from mispmm import datasets
    from mispmm import dist as mdist
    csr = datasets.load_csr("qh1484")
    part = mdist.csr_row_slice(csr, 100, 350)
    assert part.num_rows == 250 and part.row_ptrs[0] == 0 and part.row_ptrs[-1] == part.nnz
    assert np.array_equal(part.to_dense(), csr.to_dense()[100:350])
    empty = mdist.csr_row_slice(csr, 7, 7)
    assert empty.num_rows == 0 and empty.nnz == 0


def test_bench_parent_stops_every_rank_when_one_dies():
    """`python bench.py --gpus 2` started plainly spawns its ranks itself and watches all of them: here (no GPU) every
    rank dies at its first device call, and the parent must come back promptly with a non-zero status and the failing
    rank's message instead of waiting on rank 0."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU (the ranks must fail)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and time.time() - t0 < 60
    assert "exited with" in p.stderr and "the other ranks were stopped" in p.stderr
    assert p.stdout.strip() == ""
