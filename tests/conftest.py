import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_DIST_GPU_JOB = None


def pytest_sessionstart(session):
    """`-m gpu` on a box with a GPU: start the two-rank job of tests/test_dist_gpu.py NOW, as child processes, while this
    process has not touched the GPU yet (counting devices does not initialise it)."""
    global _DIST_GPU_JOB
    expr = (session.config.getoption("-m") or "").strip()
    if "gpu" not in expr or "not gpu" in expr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:  # noqa: BLE001
        return
    import socket
    import subprocess
    import tempfile
    out_dir = tempfile.mkdtemp(prefix="asurf_dist_gpu_")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    log = os.path.join(out_dir, "job.log")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out_dir]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.Popen(cmd, stdout=open(log, "w"), stderr=subprocess.STDOUT, env=env, cwd=ROOT)
    _DIST_GPU_JOB = (proc, out_dir, log)


@pytest.fixture(scope="session")
def dist_gpu_job():
    if _DIST_GPU_JOB is None:
        pytest.skip("the two-rank job was not started (no GPU, or the session was not selected with -m gpu)")
    return _DIST_GPU_JOB


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with g++."""
    from oracle import oracle as O
    O.build()
    return O


_SWITCHES = ("ASURF_DEBUG", "ASURF_WALK_SEG", "ASURF_WALK", "ASURF_RPW", "ASURF_NO_PIPELINE", "ASURF_CUVOL_RPW",
             "ASURF_NO_REPLAY", "ASURF_NO_SPLIT_BWD", "ASURF_EVAL", "ASURF_MASKS", "ASURF_SWEEP", "ASURF_SIDE_STREAM", "ASURF_ORDER",
             "ASURF_LAZY", "ASURF_BWD_SH", "ASURF_TILES", "ASURF_BWD_TILE", "ASURF_BWD_SORT", "ASURF_OPTIM", "ASURF_RELAXED", "ASURF_SUM_LIST", "ASURF_CUVOL_TILE")


@pytest.fixture(params=["value_planes", "links_planes"])
def planes(request):
    """Both forms of the occupancy planes for the small batches the tests render: the per-call planes that hold the value
    gates (the default), and planes of `links` alone kept from call to call with the gates applied where a ray meets a
    voxel (asurf_masks2.h; ASURF_LAZY, off by default)."""
    from alphasurf_amd import lib
    lib.config_set("ASURF_LAZY", "32768" if request.param == "links_planes" else None)
    return request.param


@pytest.fixture(autouse=True)
def _reset_library_switches():
    """The internal switches a test sets through lib.config_set() do not outlive it (a failing test would otherwise
    leave the library in its configuration for every test that follows).  Only when the library is loaded already."""
    yield
    from alphasurf_amd import lib
    if lib._lib is not None:
        for name in _SWITCHES:
            lib.config_set(name, None)
