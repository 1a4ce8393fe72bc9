"""The C++ host layer: `cuspmm` CLI (the reference's command line) and the class API test binary.
CPU part: --cpu-only runs the sequential engines of all four formats on files written by the
reference converter and checks records and saved results.  GPU part: the full engine flow."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-optimization-for-spmm_amd")
CLI = os.path.join(PKG, "cuspmm")
API_TEST = os.path.join(PKG, "host_api_test")
REF_KEYS = ["testcase", "sparsity", "format", "kernelType", "denseOrdering", "correct", "cudaPrologTimeMs",
            "cudaKernelTimeMs", "cudaEpilogTimeMs", "cudaTotalTimeMs", "sequentialTimeMs"]


def run_cli(*args, check=True):
    p = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=600)
    if check:
        assert p.returncode == 0, p.stderr
    return p


def records(stdout):
    """The output is a sequence of `{...},` objects with quoted string values (reference format)."""
    out = []
    for body in re.findall(r"\{(.*?)\},", stdout, flags=re.S):
        pairs = re.findall(r'"([A-Za-z]+)":"([^"]*)"', body)
        out.append((dict(pairs), [k for k, _ in pairs]))
    return out


@pytest.fixture(scope="module", autouse=True)
def _built():
    assert os.path.exists(CLI) and os.path.exists(API_TEST), "run `make -C cuda-optimization-for-spmm_amd`"


@pytest.mark.parametrize("d", ["small_32x32_generated", "small_210_generated", "small_10x10_generated"])
def test_cpu_only_all_formats_match_reference_expectation(tmp_path, golden_dir, d):
    g = os.path.join(golden_dir, d)
    expect = np.loadtxt(os.path.join(g, "result.expect"), ndmin=2)
    for flag, fmt in (("--csr", "CSR"), ("--coo", "COO"), ("--bsr", "BSR"), ("--ell", "ELL")):
        out = tmp_path / f"{fmt}.txt"
        p = run_cli(flag, "--cpu-only", "-d", g, "--save", str(out))
        recs = records(p.stdout)
        assert len(recs) == 1
        rec, keys = recs[0]
        assert keys == REF_KEYS                     # same keys, same order as the reference's reportTime
        assert rec["format"] == fmt and rec["kernelType"] == "0" and rec["correct"] == "1" and rec["testcase"] == g
        got = np.loadtxt(out, skiprows=1, ndmin=2)
        assert got.shape == expect.shape
        assert np.allclose(got, expect, rtol=1e-5, atol=1e-6)


def test_cli_argument_errors(golden_dir, tmp_path):
    assert run_cli(check=False).returncode != 0                               # nothing selected: help + failure
    assert "Usage" in run_cli("--csr", check=False).stdout                    # no -d
    assert run_cli("-h").returncode == 0
    p = run_cli("--bsr", "--cpu-only", "-d", os.path.join(golden_dir, "small_32x32"), check=False)
    assert p.returncode != 0 and "Missing required files *.bsr" in p.stderr   # the reference forgets to exit here
    empty = tmp_path / "empty"
    empty.mkdir()
    p = run_cli("--csr", "--cpu-only", "-d", str(empty), check=False)
    assert p.returncode != 0 and "*.csr" in p.stderr
    p = run_cli("--csr", "--cpu-only", "-d", str(tmp_path / "nope"), check=False)
    assert p.returncode != 0
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / "a.csr").write_text("4 4 3\n0 1 2\n")
    (bad / "dense.in").write_text("4 2\n1 2\n3 4\n5 6\n7 8\n")
    p = run_cli("--csr", "--cpu-only", "-d", str(bad), check=False)
    assert p.returncode != 0 and "malformed" in p.stderr


def test_cli_synthetic_operand_matches_python_generator(tmp_path, golden_dir):
    """-k builds B with the same counter-based generator as mispmm.synth (no dense.in needed)."""
    from mispmm import formats, synth
    g = os.path.join(golden_dir, "small_32x32")
    out = tmp_path / "c.txt"
    run_cli("--csr", "--cpu-only", "-k", "8", "-d", g, "--save", str(out))
    csr = formats.read_csr(os.path.join(g, "Hamrle1.csr"))
    want = csr.to_dense().astype(np.float64) @ synth.dense_b(32, 8).astype(np.float64)
    assert np.allclose(np.loadtxt(out, skiprows=1, ndmin=2), want, rtol=1e-5, atol=1e-6)


def test_host_api_cpu(tmp_path, golden_dir):
    from mispmm import synth
    dump = tmp_path / "syn.txt"
    p = subprocess.run([API_TEST, os.path.join(golden_dir, "small_32x32_generated"), "--cpu", str(dump)],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    got = np.loadtxt(dump, skiprows=1, ndmin=2).astype(np.float32)
    assert np.array_equal(got, synth.dense_b(7, 5))      # C++ and Python generators are bit-identical


@pytest.mark.gpu
def test_host_api_gpu(golden_dir):
    p = subprocess.run([API_TEST, os.path.join(golden_dir, "small_32x32_generated"), "--gpu"], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr + p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("d", ["small_32x32_generated", "small_210_generated"])
def test_cli_full_engine_flow_on_gpu(golden_dir, d):
    g = os.path.join(golden_dir, d)
    expected_kernels = {"CSR": ["0", "1", "2", "3", "4", "5", "6", "-1"], "COO": ["0", "1", "2", "-1"], "BSR": ["0", "1", "2", "3"],
                        "ELL": ["0", "1"]}
    p = run_cli("--csr", "--coo", "--bsr", "--ell", "--iters", "20", "-d", g)
    recs = records(p.stdout)
    by_fmt = {}
    for rec, keys in recs:
        assert keys[:len(REF_KEYS)] == REF_KEYS
        by_fmt.setdefault(rec["format"], []).append(rec)
    for fmt, kernels in expected_kernels.items():
        got = [r["kernelType"] for r in by_fmt[fmt]]
        if fmt == "BSR":                      # 1x1 blocks: the MFMA kernel declines and reports zeros
            assert got == kernels and by_fmt[fmt][2]["correct"] == "0" and by_fmt[fmt][2]["cudaKernelTimeMs"] == "0.000000"
            checked = by_fmt[fmt][:2] + by_fmt[fmt][3:]          # kernel 3 = the zero-skipping one
        else:
            assert got == kernels, (fmt, got)
            checked = by_fmt[fmt]
        assert all(r["correct"] == "1" for r in checked), [(r["kernelType"], r["correct"]) for r in checked]
        assert all(float(r["gflops"]) > 0 for r in checked if r["kernelType"] not in ("0", "-1"))


@pytest.mark.gpu
def test_cli_headline_directory_from_packed_matrix(tmp_path):
    """Build data/large_25605 the way scripts/data.sh would (text files from the packed matrix),
    run `cuspmm --csr --ell -k 128`: every kernel and the rocSPARSE check agree with the CPU engine."""
    from mispmm import datasets, formats
    d = tmp_path / "large_25605"
    d.mkdir()
    csr = datasets.load_csr("n4c6-b13", dtype=np.float64)
    formats.write_csr(d / "n4c6-b13.csr", csr, integer=True)
    formats.write_ell_colmajor(d / "n4c6-b13_rowind.ell", d / "n4c6-b13_values_colmajor.ell",
                               formats.csr_to_ell_colmajor(csr, reference_width=True), integer=True)
    p = run_cli("--csr", "--ell", "-k", "128", "--iters", "100", "-d", str(d))
    recs = [r for r, _ in records(p.stdout)]
    assert [r["kernelType"] for r in recs if r["format"] == "CSR"] == ["0", "1", "2", "3", "4", "5", "6", "-1"]
    assert all(r["correct"] == "1" for r in recs)
    # (the steady-state floors of this run live in tests/test_zz_perf_gpu.py)
    assert all(float(r["rooflineFrac"]) > 0 for r in recs if "rooflineFrac" in r and r["kernelType"] not in ("0", "-1"))


@pytest.mark.gpu
def test_cli_long_row_matrix_takes_the_split_kernel(tmp_path):
    """GL7d25 (rows of up to 422 entries, sorted so that the long ones come last) through `cuspmm --csr -k 128`: every
    kernel agrees with the CPU engine; copy2Device builds the longest-first row list, kernel 5 / the default take the
    two-body launch and kernel 6 the split kernel (their rates: tests/test_zz_perf_gpu.py)."""
    from mispmm import datasets, formats
    d = tmp_path / "GL7d25"
    d.mkdir()
    formats.write_csr(d / "GL7d25.csr", datasets.load_csr("GL7d25", dtype=np.float64), integer=True)
    p = run_cli("--csr", "-k", "128", "--iters", "200", "-d", str(d))
    recs = [r for r, _ in records(p.stdout)]
    assert [r["kernelType"] for r in recs] == ["0", "1", "2", "3", "4", "5", "6", "-1"]
    assert all(r["correct"] == "1" for r in recs)
    # kernel 5 / the default: the two-body launch (kept: 5.1 us = 0.31); kernel 6 by name: the split kernel on every row
    tags = {r["kernelType"]: r["kernel"] for r in recs if "kernel" in r}
    assert "csr_hybrid" in tags.get("5", "") and "csr_split" in tags.get("6", ""), (tags, p.stdout[-1500:])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["headline", "2", "3", "4", "5"])
def test_bench_line_contract(bench_line, cfg):
    """bench.py at the driver's flags, every BASELINE configuration: the line carries the contract's fields, the GPU result
    that was timed equals the oracle's (bench.py refuses to print otherwise), `roofline` and `hbm_streaming` follow from
    their own numbers.  No assertion on a time here (tests/test_zz_perf_gpu.py holds those)."""
    line = bench_line(cfg)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "hbm_streaming"):
        assert key in line, key
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert line["cpu_baseline"]["gpu_parity"] == ("bit-exact" if cfg != "4" else "within 2e-6 of sum|a||b| (bf16-rounded inputs, fp32 accumulate)")
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["algorithmic_bytes_per_launch"] / (line["ms_per_step"] * 1e-3) / 8e12) < 2e-3
    hs = line["hbm_streaming"]
    assert hs["bytes_in_rotation"] >= 512 << 20 and hs["operand_sets"] >= 4 and hs["launches_per_graph"] % hs["operand_sets"] == 0
    assert abs(hs["frac"] - hs["algorithmic_bytes_per_launch"] / (hs["launch_us"] * 1e-6) / 8e12) < 2e-3
    assert hs["kernel_tag"] == line["config"]["kernel_tag"]
    if cfg == "4":   # the fraction is quoted against the bytes the running kernel moves, the BSR-16 operand beside it
        assert r["algorithmic_bytes_per_launch"] < 20e6 < r["bsr16_operand_bytes"] and r["frac_vs_bsr16_operand"] > r["frac"]
        assert "bsrc_slots" in line["config"]["kernel_tag"]


@pytest.mark.gpu
def test_cli_batched_launch_and_kernel_tags(tmp_path):
    """`cuspmm --csr --batch 8`: one more CSR record (key "batch") for 8 dense operands multiplied in one launch, every
    result checked; every GPU record names the device kernel that ran (key "kernel"), so records of kernel ids that share
    a kernel say so."""
    from mispmm import datasets, formats
    d = tmp_path / "large_25605"
    d.mkdir()
    formats.write_csr(d / "n4c6-b13.csr", datasets.load_csr("n4c6-b13", dtype=np.float64), integer=True)
    p = run_cli("--csr", "-k", "128", "--batch", "8", "--no-vendor", "--iters", "400", "-d", str(d))
    recs = [r for r, _ in records(p.stdout)]
    batched = [r for r in recs if "batch" in r]
    assert len(batched) == 1 and batched[0]["batch"] == "8" and batched[0]["correct"] == "1"
    assert "batched" in batched[0]["kernel"] and float(batched[0]["steadyKernelUs"]) > 0
    single = {r["kernelType"]: r for r in recs if "batch" not in r}
    assert "row_gather" in single["5"]["kernel"] and "uniform" in single["5"]["kernel"]
    assert all("kernel" in r for k, r in single.items() if k != "0")
    # (what batching buys at this size is asserted with the other timings: tests/test_zz_perf_gpu.py)


@pytest.mark.gpu
def test_cli_accepts_a_coo_file_in_any_entry_order(tmp_path):
    """The reference's COO kernel (one atomicAdd per entry) takes entries in any order; the row-walking HIP kernel
    needs them grouped by row, so SparseMatrixCOO::copy2Device puts a shuffled file into stable row order.  A
    shuffled .coo must give `correct: 1` for every kernel and the vendor check."""
    from mispmm import datasets, formats
    d = tmp_path / "shuffled"
    d.mkdir()
    csr = datasets.load_csr("qh1484", dtype=np.float64)
    coo = formats.csr_to_coo(csr)
    perm = np.random.default_rng(3).permutation(coo.nnz)
    shuffled = formats.COO(coo.num_rows, coo.num_cols, coo.row_idxs[perm], coo.col_idxs[perm], coo.data[perm])
    formats.write_coo(d / "qh1484.coo", shuffled)
    p = run_cli("--coo", "-k", "64", "-d", str(d))
    recs = [r for r, _ in records(p.stdout)]
    assert [r["kernelType"] for r in recs] == ["0", "1", "2", "-1"]
    assert all(r["correct"] == "1" for r in recs), [(r["kernelType"], r["correct"]) for r in recs]


@pytest.mark.gpu
def test_cli_vendor_check_for_bsr(tmp_path):
    """--vendor-bsr: rocSPARSE BSR SpMM (the descriptor the reference builds at sparse_bsr.cu:138-160 but never uses)
    is timed and compared like the CSR / COO checks."""
    from mispmm import datasets, formats
    d = tmp_path / "bsr4"
    d.mkdir()
    csr = datasets.load_csr("qh1484", dtype=np.float64)
    formats.write_bsr(d / "qh1484.bsr", formats.csr_to_bsr(csr, 4))
    p = run_cli("--bsr", "--vendor-bsr", "-k", "64", "-d", str(d))
    recs = [r for r, _ in records(p.stdout)]
    assert [r["kernelType"] for r in recs] == ["0", "1", "2", "3", "-1"]
    assert recs[1]["correct"] == "1" and recs[3]["correct"] == "1" and recs[4]["correct"] == "1", [(r["kernelType"], r["correct"]) for r in recs]
    # without the flag the BSR engine reports what the reference's does: no vendor record
    p = run_cli("--bsr", "-k", "64", "-d", str(d))
    assert [r["kernelType"] for r, _ in records(p.stdout)] == ["0", "1", "2", "3"]


def test_validate_tool_checks_cli_dumps(tmp_path, golden_dir):
    """tools/validate.py (the reference's validate.py role): expected product from result.expect, *.out dumps
    written by `cuspmm --save` compared against it; a corrupted dump is reported and fails the run."""
    import shutil
    import sys
    d = tmp_path / "small_32x32"
    shutil.copytree(os.path.join(golden_dir, "small_32x32"), d)
    run_cli("--csr", "--cpu-only", "-d", str(d), "--save", str(d / "csr_cpu.out"))
    run_cli("--coo", "--cpu-only", "-d", str(d), "--save", str(d / "coo_cpu.out"))
    tool = os.path.join(ROOT, "tools", "validate.py")
    p = subprocess.run([sys.executable, tool, str(d), "--rtol", "1e-5", "--atol", "1e-6"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("matches the expected result") == 2 and "Expect file found" in p.stdout
    # without result.expect it is computed (float64) and written with 10 decimals, like the reference's
    os.remove(d / "result.expect")
    p = subprocess.run([sys.executable, tool, str(d), "--rtol", "1e-5", "--atol", "1e-6"], capture_output=True, text=True)
    assert p.returncode == 0 and "Calculated expected result" in p.stdout
    assert open(d / "result.expect").readline() == open(os.path.join(golden_dir, "small_32x32", "result.expect")).readline()
    bad = np.loadtxt(d / "csr_cpu.out", skiprows=1, ndmin=2)
    bad[3, 4] += 1.0
    np.savetxt(d / "broken.out", bad)
    p = subprocess.run([sys.executable, tool, str(d), "--rtol", "1e-5", "--atol", "1e-6"], capture_output=True, text=True)
    assert p.returncode == 1 and "broken.out does NOT match" in p.stdout


@pytest.mark.gpu
def test_cli_row_sharded_run_behind_the_gpus_flag(tmp_path):
    """`cuspmm --csr -k 512 --gpus 1` (BASELINE config 5's shape on the one card of the test box): the row-sharded run
    goes through mispmm_multi_csr_f32 and adds one record carrying `ngpus`; every gather mode agrees with the CPU engine."""
    from mispmm import datasets, formats
    d = tmp_path / "large_25605"
    d.mkdir()
    formats.write_csr(d / "n4c6-b13.csr", datasets.load_csr("n4c6-b13", dtype=np.float64), integer=True)
    for gather in ("first", "peer", "rccl", "none"):
        p = run_cli("--csr", "-k", "512", "--gpus", "1", "--gather", gather, "--no-vendor", "--iters", "20", "-d", str(d))
        recs = [r for r, _ in records(p.stdout)]
        multi = [r for r in recs if "ngpus" in r]
        assert len(multi) == 1 and multi[0]["ngpus"] == "1" and multi[0]["correct"] == "1", (gather, multi)
        assert float(multi[0]["gflops"]) > 0
        assert [r["kernelType"] for r in recs if "ngpus" not in r] == ["0", "1", "2", "3", "4", "5", "6"]
    p = run_cli("--csr", "-k", "8", "--gpus", "99", "-d", str(d), check=False)
    assert p.returncode != 0 and "--gpus 99" in p.stderr


@pytest.mark.gpu
def test_cli_ell_sharded_by_rows_behind_the_gpus_flag(tmp_path):
    """`cuspmm --ell -k 256 --gpus 1` (BASELINE config 3's operand on the one card of the test box): the ELL product goes
    through mispmm_multi_ell_f32 (rows cut by occupied slots) and adds one record carrying `ngpus`; every gather mode agrees
    with the CPU engine."""
    from mispmm import datasets, formats
    d = tmp_path / "large_25605"
    d.mkdir()
    csr = datasets.load_csr("n4c6-b13", dtype=np.float64)
    formats.write_ell_colmajor(d / "n4c6-b13_rowind.ell", d / "n4c6-b13_values_colmajor.ell",
                               formats.csr_to_ell_colmajor(csr, reference_width=True), integer=True)
    for gather in ("first", "peer", "rccl", "none"):
        p = run_cli("--ell", "-k", "256", "--gpus", "1", "--gather", gather, "--iters", "20", "-d", str(d))
        recs = [r for r, _ in records(p.stdout)]
        multi = [r for r in recs if "ngpus" in r]
        assert len(multi) == 1 and multi[0]["ngpus"] == "1" and multi[0]["correct"] == "1" and multi[0]["format"] == "ELL", (gather, multi)
        assert float(multi[0]["gflops"]) > 0
        assert [r["kernelType"] for r in recs if "ngpus" not in r] == ["0", "1"]


@pytest.mark.gpu
def test_cli_bf16_block_products_behind_the_dtype_flag(tmp_path):
    """`cuspmm --bsr --dtype bf16` (BASELINE config 4 through the CLI): the three bf16 MFMA kernels run after the fp32 ones,
    are checked against the sequential engine on bf16-rounded operands, and their records carry "dtype":"bf16"."""
    from mispmm import datasets, formats
    d = tmp_path / "medium_2048"
    d.mkdir()
    formats.write_bsr(d / "dw1024.bsr", formats.csr_to_bsr(datasets.load_csr("dw1024", dtype=np.float64), 16))
    p = run_cli("--bsr", "--dtype", "bf16", "-k", "128", "--iters", "20", "-d", str(d))
    recs = [r for r, _ in records(p.stdout)]
    assert [r["kernelType"] for r in recs] == ["0", "1", "2", "3", "6", "4", "5"]
    assert all(r["correct"] == "1" for r in recs), [(r["kernelType"], r["correct"]) for r in recs]
    assert [r.get("dtype") for r in recs] == [None, None, None, None, "bf16", "bf16", "bf16"]
    assert all(float(r["gflops"]) > 0 for r in recs[4:])
    assert "bsrc_slots" in recs[4]["kernel"] and "bsrc_mfma" in recs[5]["kernel"] and "bsr_mfma_bf16" in recs[6]["kernel"]
    p = run_cli("--bsr", "--dtype", "fp8", "-d", str(d), check=False)
    assert p.returncode != 0 and "--dtype" in p.stderr
