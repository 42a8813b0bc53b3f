"""MNIST loader (IDX parsing, synthetic fallback, batching) and the step-time reporter."""
import gzip
import struct

import numpy as np

from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.utils.step_timer import StepTimeReporter


def _write_idx(path, arr, gz):
    arr = np.asarray(arr, dtype=np.uint8)
    header = struct.pack(">HBB", 0, 0x08, arr.ndim) + b"".join(struct.pack(">I", d) for d in arr.shape)
    opener = gzip.open if gz else open
    with opener(path, "wb") as fh:
        fh.write(header + arr.tobytes())


def test_idx_files_are_read_and_split(tmp_path):
    rng = np.random.RandomState(0)
    tr_x = rng.randint(0, 256, size=(6000, 28, 28)); tr_y = rng.randint(0, 10, size=(6000,))
    te_x = rng.randint(0, 256, size=(100, 28, 28)); te_y = rng.randint(0, 10, size=(100,))
    _write_idx(str(tmp_path / "train-images-idx3-ubyte.gz"), tr_x, True)
    _write_idx(str(tmp_path / "train-labels-idx1-ubyte.gz"), tr_y, True)
    _write_idx(str(tmp_path / "t10k-images-idx3-ubyte"), te_x, False)
    _write_idx(str(tmp_path / "t10k-labels-idx1-ubyte"), te_y, False)
    ds = mnist.read_data_sets(str(tmp_path), one_hot=True, validation_size=1000)
    assert ds.train.num_examples == 5000 and ds.validation.num_examples == 1000
    assert ds.test.images.shape == (100, 784) and ds.test.labels.shape == (100, 10)
    assert ds.train.images.dtype == np.float32 and 0.0 <= ds.train.images.min() and ds.train.images.max() <= 1.0
    np.testing.assert_allclose(ds.test.images[3], te_x[3].reshape(-1) / 255.0, rtol=1e-6)
    assert ds.test.labels[5].argmax() == te_y[5] and ds.test.labels[5].sum() == 1.0
    assert not ds.synthetic


def test_synthetic_fallback_has_mnist_shapes(tmp_path):
    ds = mnist.read_data_sets(str(tmp_path / "missing"), one_hot=True)
    assert ds.synthetic
    assert ds.train.num_examples == 55000 and ds.validation.num_examples == 5000
    assert ds.test.num_examples == 10000
    assert ds.train.images.shape == (55000, 784) and ds.train.labels.shape == (55000, 10)
    # deterministic across processes/tasks
    ds2 = mnist.read_data_sets(str(tmp_path / "missing"), one_hot=True)
    np.testing.assert_array_equal(ds.test.images[:10], ds2.test.images[:10])


def test_next_batch_epochs_and_shuffling():
    x = np.arange(10, dtype=np.float32).reshape(10, 1)
    y = np.arange(10, dtype=np.float32).reshape(10, 1)
    ds = mnist.DataSet(x, y, seed=0)
    seen = []
    for _ in range(5):
        bx, by = ds.next_batch(4)
        assert bx.shape == (4, 1)
        np.testing.assert_array_equal(bx, by)           # images and labels stay aligned
        seen += bx.reshape(-1).tolist()
    assert ds.epochs_completed == 1                     # the 2nd epoch ends exactly at sample 20, not past it
    assert sorted(seen[:10]) == list(range(10))          # first epoch visits everything once


def test_step_time_reporter_windows(capsys):
    clock = [0.0]
    rep = StepTimeReporter(frequency=4, clock=lambda: clock[0])
    rep.start_epoch()
    out = []
    for i in range(10):
        clock[0] += 0.002
        out.append(rep.step(i, 0, i, 10, 1.5))
    # a line at every 4th step and at the end of the epoch, each averaged over ITS window
    assert [round(v, 3) if v else None for v in out] == [None, None, None, 2.0, None, None, None, 2.0, None, 2.0]
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 3
    assert lines[0].startswith("Step: 4,") and "Epoch:  1," in lines[0] and "Batch:   4 of  10," in lines[0]
    assert "Cost: 1.5000," in lines[0] and lines[0].rstrip().endswith("AvgTime: 2.00ms")


def test_step_reports_mirrored_into_jsonl_sink(tmp_path):
    """Every printed report is also one JSON line (flushed as written); common fields are
    merged in, numpy scalars become plain numbers, a closed sink refuses writes."""
    import numpy as np
    import pytest
    from distributed_tensorflow_amd.utils.metrics import JsonLinesSink, read_jsonl
    from distributed_tensorflow_amd.utils.step_timer import StepTimeReporter
    path = str(tmp_path / "sub" / "metrics.jsonl")
    ticks = iter(range(1000))
    lines = []
    with JsonLinesSink(path, common={"job": "worker", "task": 1}, clock=lambda: 42.0) as sink:
        rep = StepTimeReporter(100, printer=lines.append, clock=lambda: next(ticks) * 0.1,
                               sink=sink)
        rep.start_epoch()
        for i in range(550):
            rep.step(i, 0, i, 550, np.float32(2.5))
        sink.write("epoch", epoch=1, test_accuracy=np.float32(0.5), values=np.arange(3))
        got = list(read_jsonl(path))            # readable before close: line buffered
        assert len(got) == 7
    with pytest.raises(ValueError):
        sink.write("late")
    windows = [r for r in got if r["kind"] == "step_window"]
    assert len(windows) == len(lines) == 6
    assert [w["steps_in_window"] for w in windows] == [100] * 5 + [50]
    assert windows[-1]["batch"] == 550 and windows[0]["global_step"] == 100
    assert all(w["job"] == "worker" and w["task"] == 1 and w["wall_time"] == 42.0
               for w in windows)
    assert windows[0]["cost"] == 2.5
    assert got[-1] == {"kind": "epoch", "wall_time": 42.0, "job": "worker", "task": 1,
                       "epoch": 1, "test_accuracy": 0.5, "values": [0, 1, 2]}
