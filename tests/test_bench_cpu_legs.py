"""bench.py's `cpu_baseline` legs (the reference's compiled kernels, or the oracle's restatement, on host cores): every leg must
return a well-formed block on a small URM within its budget -- they run on the GPU box's host at the end of every bench, where a
failure would cost the whole JSON line."""
import numpy as np

import bench
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm


def _check(block, unit):
    assert block["unit"] == unit and block["cores"] == 1 and block["kind"] in ("reference", "port")
    assert np.isfinite(block["value"]) and block["value"] > 0 and isinstance(block["sample"], str) and block["sample"]


def test_cpu_baseline_legs_on_a_small_urm(monkeypatch, capsys):
    X = named_urm("ml1m", "binary", scale=0.25)
    monkeypatch.setattr(bench, "K_FACTORS", 16)
    monkeypatch.setattr(bench, "BATCH", 100)
    _check(bench.cpu_baseline_bpr(X, 0.3), "samples/s")
    _check(bench.cpu_baseline_funk(X, 0.3), "samples/s")
    _check(bench.cpu_baseline_slim(X, False, 0.3), "samples/s")
    _check(bench.cpu_baseline_slim(X, True, 0.3), "samples/s")
    _check(bench.cpu_baseline_asy(named_urm("ml1m", "real", scale=0.15), 8, 0.3), "samples/s")
    conf = X.copy()
    conf.data = (1.0 + conf.data).astype(np.float32)
    block = bench.cpu_baseline_ials(conf, 24, 1e-3, np.random.default_rng(0).random((X.shape[1], 24)), 0.3)
    _check(block, "s/epoch")
    out = capsys.readouterr().out
    assert "Deallocating" not in out, "the reference's prints must stay out of bench.py's stdout (one JSON line)"


def test_cpu_baseline_ials_sample_is_a_share_of_the_epochs_cost():
    """With a generous budget the leg walks its random rows until the shorter side runs out: the stated share of the epoch's flops
    is then substantial and the extrapolated epoch is the measured time divided by that share."""
    X = named_urm("ml1m", "binary", scale=0.1)
    conf = X.copy()
    conf.data = (1.0 + conf.data).astype(np.float32)
    block = bench.cpu_baseline_ials(conf, 8, 1e-3, np.random.default_rng(0).random((X.shape[1], 8)), 30.0)
    share = float(block["sample"].split("=")[1].split("%")[0])
    assert 30.0 < share <= 100.0 and block["value"] > 0
