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
    expected_kernels = {"CSR": ["0", "1", "2", "3", "4", "5", "-1"], "COO": ["0", "1", "2", "-1"], "BSR": ["0", "1", "2"],
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
            checked = by_fmt[fmt][:2]
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
    assert [r["kernelType"] for r in recs if r["format"] == "CSR"] == ["0", "1", "2", "3", "4", "5", "-1"]
    assert all(r["correct"] == "1" for r in recs)
    best = max(float(r["rooflineFrac"]) for r in recs if "rooflineFrac" in r)
    assert best > 0.2, "steady-state HBM roofline fraction collapsed"


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
