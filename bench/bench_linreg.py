#!/usr/bin/env python3
"""Step latency of the linear-regression SGD step (1024 features, batch 256) through
``Session.run`` on one GPU: fused mean-squared-error op against the composition
``reduce_mean(square(y - y_))``, waiting for every step's results and with deferred fetches."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.models import linear_regression as lr  # noqa: E402


def run(fused: bool, deferred: bool, steps: int = 3000, warmup: int = 100) -> dict:
    with tf.Graph().as_default():
        m = lr.build_model(fused_loss=fused)
        g = torch.Generator().manual_seed(0)
        xs = torch.randn(16, 256, 1024, generator=g).cuda()
        ys = torch.randn(16, 256, 1, generator=g).cuda()
        cfg = tf.ConfigProto(device="cuda:0", deferred_fetches=deferred)
        with tf.Session(config=cfg) as sess:
            sess.run(m.init_op)
            fetches = [m.train_op, m.loss, m.global_step]
            for i in range(warmup):
                sess.run(fetches, {m.x: xs[i % 16], m.y_: ys[i % 16]})
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                out = sess.run(fetches, {m.x: xs[i % 16], m.y_: ys[i % 16]})
            loss = float(out[1])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / steps
    return {"bench": "linear_regression_step", "fused_loss": fused, "deferred_fetches": deferred,
            "ms_per_step": round(ms, 4), "final_loss": loss}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=("fused", "composed"), default=None,
                    help="one variant, results awaited every step (for a kernel trace)")
    ap.add_argument("--steps", type=int, default=3000)
    args = ap.parse_args()
    if args.only:
        print(json.dumps(run(args.only == "fused", False, steps=args.steps)), flush=True)
    else:
        for deferred in (True, False):
            for fused in (True, False):
                print(json.dumps(run(fused, deferred, steps=args.steps)), flush=True)
