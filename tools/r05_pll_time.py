#!/usr/bin/env python3
"""tools only (round 5): step time of the per-output objective (mll_type = "PLL" / per-output ELBO) at C2 / C3 geometry: one C call
(dsvgp_elbo_step_po_f32) against the Python-orchestrated path, and the ELBO fast path beside them"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import dsvgp_amd
from test_gpu_step import make_problem
dev = torch.device("cuda", 0)
for name, (N, d, M, p, B) in (("c2", (10000, 5, 200, 2, 512)), ("c3", (50000, 10, 300, 10, 512))):
    P, x, y, D, nd = make_problem(N, d, M, p, B, seed=1)
    Pg = {k: v.to(dev) for k, v in P.items()}
    xd, yd, Dd = x.to(dev), y.to(dev), D.to(dev)
    for mll, fast, c_step in (("ELBO", True, True), ("PLL", False, True), ("PLL", False, False), ("ELBO", False, True), ("ELBO", False, False)):
        eng = dsvgp_amd.ElboEngine(dev)
        eng.c_step = c_step
        for _ in range(5):
            eng.loss_and_grads(Pg, xd, yd, Dd, nd, mll, fast=fast)
        torch.cuda.synchronize()
        n = 100 if name == "c2" else 30
        t0 = time.perf_counter()
        for _ in range(n):
            loss, _, _, _ = eng.loss_and_grads(Pg, xd, yd, Dd, nd, mll, fast=fast)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print("%s %-4s %-10s %-22s %.3f ms/step (loss %.5f)" % (name, mll, "gram" if fast else "per-output",
                                                               "one C call" if eng.c_step_used else "Python-orchestrated", ms, loss.item()))
