"""Two ranks on one device (`pytest -m gpu`): each renders its shard of one view with the GLOBAL ray count in the loss
normaliser, the gradient buffers are exchanged (dense ring, union-of-rows, auto, and the step in two parts with the first
part's rows exchanged beside the second part's render), and every rank must end up with the
single-batch gradients and masks.  The ranks are child processes that tests/conftest.py starts at session start --
before this process touches the GPU (a process that has initialised the GPU must not start programs)."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_share_one_device(dist_gpu_job):
    proc, out_dir, log = dist_gpu_job
    rc = proc.wait(timeout=600)
    logs = open(log).read()[-3000:] if os.path.exists(log) else ""
    assert rc == 0, f"the two-rank job failed ({rc}):\n{logs}"
    for rank in range(2):
        res = json.load(open(os.path.join(out_dir, f"rank{rank}.json")))
        assert res["ok"], res.get("error")
        assert set(res["cases"]) == {"dense", "sparse", "auto", "pipelined"}
        for mode, c in res["cases"].items():
            assert c["rgb_equal"] and c["masks_equal"], (rank, mode, c)
            assert 0.01 < c["touched"] < 0.5
            for k in ("grad_sh", "grad_density", "grad_surface", "grad_fake_sample_std"):
                assert c[k] <= 2e-4, (rank, mode, k, c[k])
        assert res["cases"]["dense"]["used"] == "dense"
        assert res["cases"]["sparse"]["used"] == "sparse"
        assert res["cases"]["auto"]["used"] == "sparse"     # a view touches well under half of the rows
        assert res["cases"]["pipelined"]["used"] == "pipelined"   # the step in two parts (dist.PipelinedReduce)
