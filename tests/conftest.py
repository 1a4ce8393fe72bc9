import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-optimization-for-spmm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The order in which the files of `pytest -m gpu -x` run.  Parity first -- the curated HIP-vs-oracle suite, then the fuzz
# suite, then the sharded drivers, the CLI and the tools -- and every assertion on a TIME last (tests/test_zz_perf_gpu.py,
# marker `perf`): a perf wobble on a shared box may fail its own test but can never again stand in front of a parity test
# (round 3: a timing bound in test_gpu_multi.py stopped the driver's `-x` run before 192 parity tests).
FILE_ORDER = ["test_oracle.py", "test_formats.py", "test_capi_cpu.py", "test_dist_cpu.py",
              "test_gpu_spmm.py", "test_gpu_fuzz.py", "test_gpu_multi.py", "test_cli.py", "test_tools.py",
              "test_zz_perf_gpu.py"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: asserts on a measured time or roofline fraction (always also `gpu`; sorted last; "
                                       "deselect with -m 'gpu and not perf')")


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.basename(str(item.fspath))
        rank = FILE_ORDER.index(name) if name in FILE_ORDER else len(FILE_ORDER) - 1
        if item.get_closest_marker("perf") is not None:      # a perf test never sorts before a parity test, wherever it lives
            rank = len(FILE_ORDER)
        return rank
    items.sort(key=key)                                       # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/liboracle.so), built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def bench_line():
    """bench_line(cfg) -> the JSON line of `bench.py --config cfg` at the driver's flags (--steps 20 --warmup 5), run once per
    session and shared by the contract test (tests/test_cli.py) and the perf floors (tests/test_zz_perf_gpu.py)."""
    import json
    import subprocess
    cache = {}

    def get(cfg):
        if cfg not in cache:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "20", "--warmup", "5",
                                "--cpu-seconds", "1", "--no-extras", "--hbm-streaming", "on"], capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-2000:]
            cache[cfg] = json.loads(p.stdout.strip().splitlines()[-1])
        return cache[cfg]
    return get
