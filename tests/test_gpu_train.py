"""End-to-end training through the drop-in harness on the GPU: the headline geometry (BASELINE config 4: d = 20, M = 500, p = 5,
B = 4096) at a reduced N, shuffled epochs with a ragged last minibatch, then ``eval_gp`` on held-out rows -- what the reference's
own test prints (/root/reference/tests/test_dsvgp.py:70-103: loss per 50 steps, test MSE and mean negative predictive density).
The full-size run (one epoch over N = 1M, 245 steps) is ``tools/train_quality.py``; its output is committed under ``profiles/``."""
import importlib.util
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("train_quality", os.path.join(ROOT, "tools", "train_quality.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_c4_geometry_trains_and_predicts(dsvgp, gpu_device):
    tq = _tool()
    # 60k rows = 14 full minibatches + a ragged one of 2656 rows per epoch; 4 epochs = 60 steps (two report steps)
    args = tq.parse(["--config", "c4", "--n-train", "60000", "--n-test", "8192", "--epochs", "4", "--data", "welch"])
    res = tq.run(args)
    traj = res["trajectory"]
    assert res["total_steps"] == 60 and res["ragged_tail_rows"] == 60000 % 4096
    losses = [t["loss"] for t in traj]
    assert all(math.isfinite(v) for v in losses), losses
    # monotone-ish: the loss at step 50 and at the end is well below the first step's; no blow-up in between
    assert losses[1] < losses[0] and losses[-1] < losses[0], losses
    assert max(losses) <= losses[0] + 1e-6, losses
    assert all(math.isfinite(t["batch_nll"]) for t in traj if "batch_nll" in t)
    assert res["variances_finite"] and res["variance_min"] > 0
    assert math.isfinite(res["test_nll"]) and math.isfinite(res["test_mse"])
    # the function values are predicted better than by the training mean (standardised targets: a constant predictor scores ~1)
    assert res["test_mse"] < 0.8 * res["test_mse_of_a_constant_predictor"], res
    print("[train] c4 geometry, N = 60k, 60 steps: loss %.4f -> %.4f, test MSE %.4f (constant %.4f), NLL %.4f"
          % (losses[0], losses[-1], res["test_mse"], res["test_mse_of_a_constant_predictor"], res["test_nll"]))


def test_c2_trajectory_beside_the_oracle_trainer(dsvgp, gpu_device):
    """25 optimisation steps at BASELINE config 2's size from one initial state, on identical minibatches and derivative columns:
    the HIP harness (fused step + hand-written Adam) against the oracle's op sequence with torch.optim.Adam on the CPU."""
    tq = _tool()
    import bench
    cfg = dict(bench.CONFIGS["c2"])
    Xall, Yall = tq.make_data("sin", cfg["N"], cfg["d"], gpu_device, seed=0)
    r = tq.against_oracle(dsvgp.directional_vi, Xall.contiguous(), Yall.float().contiguous(), cfg, 25, 0.01)
    print("[train] c2 against the oracle trainer, 25 steps: max rel diff %.2e, at the last step %.2e; loss %.5f -> %.5f"
          % (r["max_rel_diff"], r["rel_diff_at_last_step"], r["hip_loss"][0], r["hip_loss"][-1]))
    assert r["max_rel_diff"] < 2e-4, r
    assert r["hip_loss"][-1] < r["hip_loss"][0]
