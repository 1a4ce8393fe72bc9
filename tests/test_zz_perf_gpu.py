"""Every assertion on a measured TIME or roofline fraction of the GPU suite lives here (markers `gpu` + `perf`), and this file
runs LAST (tests/conftest.py orders the collection): parity first, timings after it.  Under the driver's `pytest -m gpu -x`
a perf wobble on a shared box can therefore fail a test of this file but can never stand in front of a parity test again
(round 3: one timing bound, missed by 0.04 %, hid 192 parity tests).  The reference's own self-check is a correctness
check only (reference/src/spmm/csr/spmm_csr_k3.cu:97-99); these floors are regression guards of THIS repository.

Floors sit >= 8 % under the numbers kept under profiles/r4 (the same binary moves by up to 6 % with where its operands sit
in memory, DESIGN.md header).  `pytest -m "gpu and not perf"` runs the parity suite alone."""
import ctypes
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from mispmm import capi, datasets, formats, ops, synth  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.perf]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cuda-optimization-for-spmm_amd", "cuspmm")

# resident loop (`roofline.frac`) and HBM-streamed loop (`hbm_streaming.frac`) per BASELINE configuration: kept figure, floor
RESIDENT = {"headline": (0.605, 0.55), "2": (0.21, 0.19), "3": (0.72, 0.66), "4": (0.537, 0.49), "5": (0.63, 0.575)}
STREAMED = {"headline": (0.338, 0.30), "2": (0.16, 0.14), "3": (0.337, 0.30), "4": (0.46, 0.41), "5": (0.41, 0.36)}


def _records(stdout):
    return [dict(re.findall(r'"([A-Za-z]+)":"([^"]*)"', body)) for body in re.findall(r"\{(.*?)\},", stdout, flags=re.S)]


@pytest.mark.parametrize("cfg", ["headline", "2", "3", "4", "5"])
def test_bench_line_perf_floors(bench_line, cfg):
    line = bench_line(cfg)
    kept, floor = RESIDENT[cfg]
    assert line["roofline"]["frac"] >= floor, (cfg, kept, line["roofline"])
    assert line["hbm_streaming"]["frac"] >= STREAMED[cfg][1], (cfg, STREAMED[cfg][0], line["hbm_streaming"])
    # streaming B and C from / to HBM cannot be faster than finding them in the Infinity Cache (5 % for the noise of two loops)
    assert line["hbm_streaming"]["launch_us"] >= 0.95 * line["roofline"]["launch_us"], (cfg, line["hbm_streaming"], line["roofline"])


def test_cli_steady_state_floors(tmp_path):
    """`cuspmm --csr --ell -k 128 --iters 100` on the headline directory: the CLI replays its launches from one hipGraph; a
    100-node graph still carries the ~7 us of one graph launch, so the floors sit under bench.py's (kept 0.52-0.56)."""
    d = tmp_path / "large_25605"
    d.mkdir()
    csr = datasets.load_csr("n4c6-b13", dtype=np.float64)
    formats.write_csr(d / "n4c6-b13.csr", csr, integer=True)
    formats.write_ell_colmajor(d / "n4c6-b13_rowind.ell", d / "n4c6-b13_values_colmajor.ell",
                               formats.csr_to_ell_colmajor(csr, reference_width=True), integer=True)
    p = subprocess.run([CLI, "--csr", "--ell", "-k", "128", "--iters", "100", "-d", str(d)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    best = {}
    for r in _records(p.stdout):
        if "rooflineFrac" in r and r["kernelType"] not in ("0", "-1"):
            best[r["format"]] = max(best.get(r["format"], 0.0), float(r["rooflineFrac"]))
    assert best["CSR"] >= 0.44 and best["ELL"] >= 0.44, best


def test_cli_long_row_floors(tmp_path):
    """GL7d25 through the CLI: the two-body launch (kernel 5, kept 5.1 us = 0.31) and the split kernel (kernel 6, 6.9 us =
    0.23) beat the wave-per-row kernels 1-4."""
    d = tmp_path / "GL7d25"
    d.mkdir()
    formats.write_csr(d / "GL7d25.csr", datasets.load_csr("GL7d25", dtype=np.float64), integer=True)
    p = subprocess.run([CLI, "--csr", "-k", "128", "--iters", "200", "-d", str(d)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    frac = {r["kernelType"]: float(r["rooflineFrac"]) for r in _records(p.stdout) if "rooflineFrac" in r}
    assert frac["6"] >= 0.165 and frac["5"] >= 0.23, frac
    assert max(frac[k] for k in ("1", "2", "3", "4")) < frac["6"], frac


def test_cli_batched_launch_is_not_slower_per_product(tmp_path):
    """`cuspmm --csr --batch 8`: the launch boundary once per 8 products (kept 3.05-3.41 us per product against 3.38-3.61 for
    single launches; 8 operands sit in 8 places, so the bound allows the placement spread)."""
    d = tmp_path / "large_25605"
    d.mkdir()
    formats.write_csr(d / "n4c6-b13.csr", datasets.load_csr("n4c6-b13", dtype=np.float64), integer=True)
    p = subprocess.run([CLI, "--csr", "-k", "128", "--batch", "8", "--no-vendor", "--iters", "400", "-d", str(d)], capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    recs = _records(p.stdout)
    batched = [r for r in recs if "batch" in r][0]
    single = {r["kernelType"]: r for r in recs if "batch" not in r}
    assert float(batched["steadyKernelUs"]) < 1.12 * float(single["5"]["steadyKernelUs"])


def _graph_of(stream, launches, fn):
    l = capi.lib()
    sp = ctypes.c_void_p(stream.cuda_stream)
    capi.check(l.mispmm_graph_begin(sp))
    for _ in range(launches):
        fn()
    g = ctypes.c_void_p()
    capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
    return g


def _time_per_launch(run, launches, seconds=0.05):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or reps < 5:
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        reps += 5
    return (time.perf_counter() - t0) / (reps * launches)


def test_sharded_driver_kernel_only_step_is_the_plain_step():
    """ONE process, one set of operands: the kernel-only steps of the one-process-per-GPU driver (world size 1, RCCL backend,
    a bucket graph of 500 launches into its resident slab) against 500 plain launches of the same kernel on the same A and B
    into a C of their own, both replayed back to back and timed alternately.  Same kernel, same bytes, same cache residency:
    the ratio is 1 within the noise of where the two C buffers sit (fresh C alone: 1.6 %, profiles/r3/placement_probe.log)."""
    import torch.distributed as dist
    from mispmm import dist as mdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        csr = datasets.load_csr("n4c6-b13")
        n, launches = 128, 500
        job = mdist.ShardedCsrSpmm(csr, n, device=torch.device("cuda", 0), bucket=launches)
        job.broadcast_b(synth.dense_b(csr.num_cols, n))
        a = ops.DeviceCSR.from_host(csr)
        c = torch.empty((csr.num_rows, n), dtype=torch.float32, device="cuda")
        stream = job.compute_stream
        g = _graph_of(stream, launches, lambda: ops.spmm_csr(a, job.b, out=c, stream=stream))
        sp = ctypes.c_void_p(stream.cuda_stream)
        plain, sharded = [], []
        for _ in range(3):
            plain.append(_time_per_launch(lambda: capi.check(capi.lib().mispmm_graph_launch(g, sp)), launches))
            sharded.append(_time_per_launch(lambda: job.run(launches, gather=False), launches))
        job.finish(gather=False)
        assert torch.equal(job.local_slab(), c)
        ratio = min(sharded) / min(plain)
        assert 0.92 <= ratio <= 1.08, (plain, sharded)
        capi.check(capi.lib().mispmm_graph_destroy(g))
        job.close()
    finally:
        dist.destroy_process_group()
