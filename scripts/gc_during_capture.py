"""What a cyclic collection inside a stream capture does, with and without the session's guard.

A child process per case (an abort must not take the reporter down): gc threshold 1, and a dropped
instantiated HIP graph in a reference cycle left behind right as every capture begins. With
``DTF_NO_GC_PAUSE=1`` the session's guard is a no-op, the collector runs inside the capture and
finalises the old graph there; with the guard the same program finishes. Prints one JSON line per
case: exit status, tail of stderr, and how many collections began while the stream was capturing.

    python scripts/gc_during_capture.py > gpurun_out/gc_during_capture.jsonl   (LAST step of a call)
"""
import json
import os
import subprocess
import sys

CHILD = r'''
import gc, json, sys
import numpy as np, torch
import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.client import session as S

inside = [0]
def hook(phase, info):
    if phase == "start" and torch.cuda.is_current_stream_capturing():
        inside[0] += 1
gc.callbacks.append(hook)

import contextlib
old_graphs = []
t = torch.zeros(64, device="cuda")
for _ in range(4):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        t += 1
    old_graphs.append(g)
torch.cuda.synchronize()
del g
real_guard = S._gc_paused

@contextlib.contextmanager
def guard_then_garbage():
    # entered by the session right before it begins a capture: drop one old graph in a cycle
    with real_guard():
        if old_graphs:
            d = S._CapturedStep(); d.graph = old_graphs.pop(); d.cycle = d
            del d
        yield
S._gc_paused = guard_then_garbage

with tf.Graph().as_default():
    x = tf.placeholder(tf.float32, [None, 32])
    w = tf.Variable(tf.ones([32, 8]))
    step = tf.train.GradientDescentOptimizer(0.01).minimize(
        tf.reduce_mean(tf.square(tf.matmul(x, w))))
    with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
        sess.run(tf.global_variables_initializer())
        for rows in (16, 24, 40, 56):                # four shapes, four captures
            gc.set_threshold(1)
            for _ in range(6):
                sess.run(step, {x: np.ones((rows, 32), np.float32)})
            gc.set_threshold(700, 10, 10)
        print(json.dumps({"replays": sess.stats["graph_replays"],
                          "collections_inside_capture": inside[0]}))
'''

for no_guard in ("", "1"):
    env = dict(os.environ, DTF_NO_GC_PAUSE=no_guard)
    if not no_guard:
        env.pop("DTF_NO_GC_PAUSE")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True,
                       timeout=240)
    print(json.dumps({"guard": not no_guard, "exit": p.returncode,
                      "stdout": p.stdout.strip().splitlines()[-1:] or None,
                      "stderr_tail": p.stderr.strip().splitlines()[-6:]}), flush=True)
