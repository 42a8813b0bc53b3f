"""T0: CRC-32C, TFRecord event files, SSTable, Saver-V2 bundles, checkpoint state, Philox."""
import os
import struct

import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd import summary as S
from distributed_tensorflow_amd.ops import philox
from distributed_tensorflow_amd.train import saver as saver_lib
from distributed_tensorflow_amd.utils import crc32c, native, sstable


def test_crc32c_known_answers():
    # RFC 3720 / iSCSI test vectors
    vectors = [(b"", 0x00000000), (b"123456789", 0xE3069283), (b"\x00" * 32, 0x8A9136AA),
               (b"\xff" * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E),
               (bytes(range(31, -1, -1)), 0x113FDB5C)]
    for data, want in vectors:
        assert crc32c.crc32c_py(data) == want
        assert crc32c.crc32c(data) == want
    blob = os.urandom(10007)
    assert crc32c.crc32c(blob) == crc32c.crc32c_py(blob)
    assert crc32c.crc32c(blob[5000:], crc32c.crc32c(blob[:5000])) == crc32c.crc32c(blob)
    c = crc32c.crc32c(b"123456789")
    assert crc32c.mask(c) == ((((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF)
    assert crc32c.unmask(crc32c.mask(c)) == c


def test_native_host_library_is_built_and_used():
    assert native.available(), "libdtf_host.so missing: run python build_native.py"
    assert native.load().dtf_crc32c_hw() in (0, 1)


def test_tfrecord_framing_and_event_roundtrip(tmp_path):
    payload = S.encode_event(12.5, step=7, summary=S.encode_scalar_summary("cost", 1.25))
    rec = S.frame_record(payload)
    (length,) = struct.unpack("<Q", rec[:8])
    assert length == len(payload) and len(rec) == len(payload) + 16
    assert struct.unpack("<I", rec[8:12])[0] == crc32c.masked_crc32c(rec[:8])
    ev = S.decode_event(payload)
    assert ev["step"] == 7 and ev["wall_time"] == 12.5 and ev["summary"] == [("cost", 1.25)]
    # golden bytes of Summary{value{tag:"cost" simple_value:1.25}}
    assert S.encode_scalar_summary("cost", 1.25) == bytes.fromhex("0a0b0a04636f7374150000a03f")


@pytest.mark.parametrize("use_native", [True, False])
def test_file_writer_events(tmp_path, monkeypatch, use_native):
    if not use_native:
        monkeypatch.setattr(native, "_LIB", None)
        monkeypatch.setattr(native, "_TRIED", True)
    ce = tf.placeholder(tf.float32, [])
    acc = tf.placeholder(tf.float32, [])
    tf.summary.scalar("cost", ce)
    tf.summary.scalar("accuracy", acc)
    merged = tf.summary.merge_all()
    with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
        writer = tf.summary.FileWriter(str(tmp_path), graph=tf.get_default_graph())
        for step in range(25):
            s = sess.run(merged, {ce: 1.0 / (step + 1), acc: step / 25.0})
            writer.add_summary(s, step)
        writer.add_summary(None, 99)          # a None summary op fetches None: ignored
        writer.close()
    files = [f for f in os.listdir(tmp_path) if f.startswith("events.out.tfevents.")]
    assert len(files) == 1
    events = S.read_events(os.path.join(tmp_path, files[0]))
    assert events[0]["file_version"] == "brain.Event:2"
    assert b"ScalarSummary" in events[1]["graph_def"]
    scalars = [e for e in events if "summary" in e]
    assert len(scalars) == 25 and [e["step"] for e in scalars] == list(range(25))
    assert scalars[4]["summary"][0] == ("cost", pytest.approx(0.2))
    assert scalars[4]["summary"][1][0] == "accuracy"
    # corrupting one payload byte is detected
    path = os.path.join(tmp_path, files[0])
    data = bytearray(open(path, "rb").read())
    data[-10] ^= 0xFF
    open(path, "wb").write(bytes(data))
    with pytest.raises(IOError):
        S.read_events(path)


def test_file_writer_encodes_deferred_summaries_when_it_writes_them(tmp_path):
    """A summary / step that are still ``DeferredValue``s when ``add_summary`` is called are not
    waited for there: the event is put together when the writer drains its queue, in order with
    the events around it."""
    from distributed_tensorflow_amd.client.deferred import DeferredValue
    resolved = []

    def late(value, tag):
        def thunk():
            resolved.append(tag)
            return value
        return DeferredValue(thunk)

    writer = tf.summary.FileWriter(str(tmp_path))
    writer.add_summary(S.encode_scalar_summary("cost", 3.0), 1)
    writer.add_summary(late(S.encode_scalar_summary("cost", 2.0), "summary"),
                       late(np.float32(2.0), "step"))
    assert resolved == []                         # nothing looked at the values yet
    writer.add_summary(S.encode_scalar_summary("cost", 1.0), 3)
    writer.close()
    assert sorted(resolved) == ["step", "summary"]
    files = [f for f in os.listdir(tmp_path) if f.startswith("events.out.tfevents.")]
    events = [e for e in S.read_events(os.path.join(tmp_path, files[0])) if "summary" in e]
    assert [e["step"] for e in events] == [1, 2, 3]
    assert [e["summary"][0][1] for e in events] == [pytest.approx(3.0), pytest.approx(2.0),
                                                    pytest.approx(1.0)]


def test_merge_all_none_without_summaries():
    assert tf.summary.merge_all() is None


def test_sstable_golden_single_key_and_roundtrip(tmp_path):
    # one entry ("k" -> "v"): block = [shared=0, non_shared=1, vlen=1, 'k', 'v'] + restart[0] + count 1
    blob = sstable.build_table_py([(b"k", b"v")])
    block = bytes([0, 1, 1]) + b"kv" + struct.pack("<II", 0, 1)
    assert blob[:len(block)] == block and blob[len(block)] == 0
    assert struct.unpack("<I", blob[len(block) + 1:len(block) + 5])[0] == \
        crc32c.mask(crc32c.crc32c(block + b"\x00"))
    assert struct.unpack("<Q", blob[-8:])[0] == 0xDB4775248B80FB57 and len(blob[-48:]) == 48
    assert sstable.parse_table_py(blob) == [(b"k", b"v")]
    rng = np.random.RandomState(0)
    entries = [(b"", b"header")] + [(("w/%06d" % i).encode(), rng.bytes(rng.randint(0, 200)))
                                    for i in range(2000)]
    path = str(tmp_path / "t.index")
    sstable.write_table(path, entries)                 # native writer
    assert open(path, "rb").read() == sstable.build_table_py(entries)   # == python writer
    assert sstable.read_table(path) == entries
    with pytest.raises(ValueError):
        sstable.write_table(str(tmp_path / "bad"), [(b"b", b""), (b"a", b"")])
    data = bytearray(open(path, "rb").read())
    data[100] ^= 1
    open(path, "wb").write(bytes(data))
    with pytest.raises(IOError):
        sstable.read_table(path)


def test_bundle_layout_and_roundtrip(tmp_path):
    tensors = {"global_step": torch.tensor(42.0),
               "weights/Variable": torch.randn(784, 100),
               "weights/Variable_1": torch.randn(100, 10),
               "biases/Variable": torch.zeros(100),
               "half": torch.randn(7, 3).bfloat16(),
               "ints": torch.arange(12, dtype=torch.int64).reshape(3, 4)}
    prefix = str(tmp_path / "model.ckpt-42")
    saver_lib.write_bundle(prefix, [tensors])
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    num_shards, index = saver_lib.read_bundle_index(prefix)
    assert num_shards == 1 and sorted(index) == sorted(tensors)
    e = index["weights/Variable"]
    assert e["dtype"] == 1 and e["shape"] == [784, 100] and e["size"] == 784 * 100 * 4
    # data file = tensors back to back in key order
    order = sorted(tensors)
    offsets = [index[n]["offset"] for n in order]
    assert offsets == sorted(offsets) and offsets[0] == 0
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(index[n]["size"] for n in order)
    back = saver_lib.read_bundle(prefix)
    for n, t in tensors.items():
        assert back[n].dtype == t.dtype and torch.equal(back[n], t)
    # header entry: BundleHeaderProto{num_shards=1, version{producer=1}}
    rows = sstable.read_table(prefix + ".index")
    assert rows[0] == (b"", bytes.fromhex("08011a020801"))
    with pytest.raises(KeyError):
        saver_lib.read_bundle(prefix, ["nope"])
    blob = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    blob[10] ^= 0xFF
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(blob))
    with pytest.raises(IOError):
        saver_lib.read_bundle(prefix)


def test_saver_save_restore_and_checkpoint_state(tmp_path):
    step = tf.train.get_or_create_global_step()
    w = tf.Variable(tf.random_normal([5, 3], seed=3), name="w")
    b = tf.Variable(tf.zeros([3]), name="b")
    bump = tf.group(w.assign_add(tf.ones([5, 3])), step.assign_add(1.0))
    saver = tf.train.Saver(max_to_keep=2)
    ckpt_dir = str(tmp_path / "ckpt")
    with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
        sess.run(tf.global_variables_initializer())
        paths, values = [], []
        for _ in range(3):
            sess.run(bump)
            paths.append(saver.save(sess, os.path.join(ckpt_dir, "model.ckpt"), global_step=step))
            values.append(sess.run(w).copy())
    assert [os.path.basename(p) for p in paths] == ["model.ckpt-1", "model.ckpt-2", "model.ckpt-3"]
    assert not os.path.exists(paths[0] + ".index")              # max_to_keep=2 pruned the oldest
    assert tf.train.latest_checkpoint(ckpt_dir) == paths[2]
    state = tf.train.get_checkpoint_state(ckpt_dir)
    assert state["all_model_checkpoint_paths"] == paths[1:]
    text = open(os.path.join(ckpt_dir, "checkpoint")).read()
    assert text.startswith('model_checkpoint_path: "model.ckpt-3"\n')
    with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:    # restore without init
        saver.restore(sess, paths[1])
        assert np.array_equal(sess.run(w), values[1]) and sess.run(step) == 2.0
        with pytest.raises(ValueError):
            saver.restore(sess, os.path.join(ckpt_dir, "missing"))
    assert tf.train.latest_checkpoint(str(tmp_path / "empty")) is None
    # looking into the files without a graph: by prefix and by directory (= latest checkpoint)
    assert tf.train.list_variables(paths[1]) == [("b", [3]), ("global_step", []), ("w", [5, 3])]
    assert np.array_equal(tf.train.load_variable(paths[1], "w:0"), values[1])
    reader = tf.train.NewCheckpointReader(ckpt_dir)
    assert reader.has_tensor("w") and not reader.has_tensor("nope")
    assert np.array_equal(reader.get_tensor("w"), values[2])
    assert reader.get_variable_to_dtype_map()["w"] == tf.float32
    assert "w (float32) [5, 3]" in reader.debug_string()
    with pytest.raises(KeyError):
        reader.get_tensor("nope")
    with pytest.raises(ValueError):
        tf.train.NewCheckpointReader(str(tmp_path / "empty"))


def test_sharded_bundle_two_data_files(tmp_path):
    prefix = str(tmp_path / "s.ckpt")
    saver_lib.write_bundle(prefix, [{"a": torch.ones(3)}, {"b": torch.arange(4.0)}])
    assert os.path.exists(prefix + ".data-00000-of-00002")
    assert os.path.exists(prefix + ".data-00001-of-00002")
    n, index = saver_lib.read_bundle_index(prefix)
    assert n == 2 and index["a"]["shard_id"] == 0 and index["b"]["shard_id"] == 1
    back = saver_lib.read_bundle(prefix)
    assert torch.equal(back["b"], torch.arange(4.0))
    with pytest.raises(ValueError):
        saver_lib.write_bundle(prefix, [{"a": torch.ones(1)}, {"a": torch.ones(1)}])


def test_philox_known_answer_and_stream_properties():
    # Random123 known-answer test: Philox4x32-10, counter = 0, key = 0
    out = philox.philox4x32(np.array([0], dtype=np.uint64), 0, 0)[0]
    assert [hex(int(v)) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    z = philox.normal(200001, 1, 2)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert np.array_equal(philox.normal(1000, 1, 2, offset=777), z[777:1777])   # any slice, directly
    assert not np.array_equal(philox.normal(100, 1, 3), z[:100])
    u = philox.uniform(100000, 5, 6)
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    win = philox.window(4, 6, 1, 2, row0=10, col0=3, global_cols=50)
    full = philox.normal(50 * 20, 1, 2).reshape(20, 50)
    assert np.allclose(win, full[10:14, 3:9], atol=1e-6)


def test_histogram_summary_buckets_and_event_round_trip(tmp_path):
    """``tf.summary.histogram``: TensorFlow's default bucket limits, empty runs collapsed the way
    its encoder collapses them, statistics exact, and the value survives an event file."""
    from distributed_tensorflow_amd.summary import (decode_histograms, encode_histogram_summary,
                                                    histogram_bucket_limits)
    limits = histogram_bucket_limits()
    assert np.all(np.diff(limits) > 0) and limits[len(limits) // 2] == 0.0
    assert np.isclose(limits[len(limits) // 2 + 1], 1e-12) and limits[-1] == np.finfo(np.float64).max
    (tag, h), = decode_histograms(encode_histogram_summary("one", [0.5]))
    assert tag == "one" and h["bucket"] == [0.0, 1.0, 0.0]
    assert h["bucket_limit"][0] <= 0.5 < h["bucket_limit"][1] <= 0.5 * 1.1
    assert h["bucket_limit"][2] == np.finfo(np.float64).max
    assert (h["min"], h["max"], h["num"], h["sum"], h["sum_squares"]) == (0.5, 0.5, 1.0, 0.5, 0.25)
    vals = np.random.RandomState(0).randn(5000) * 3
    (_, h), = decode_histograms(encode_histogram_summary("w", vals))
    assert sum(h["bucket"]) == 5000 and h["num"] == 5000
    assert np.isclose(h["sum"], vals.sum()) and np.isclose(h["sum_squares"], (vals ** 2).sum())
    assert all(a < b for a, b in zip(h["bucket_limit"], h["bucket_limit"][1:]))
    # no two neighbouring entries are both empty (runs are collapsed)
    assert not any(a == 0 and b == 0 for a, b in zip(h["bucket"], h["bucket"][1:]))
    # every value lies below its bucket's limit and at or above the previous one
    edges = np.array(h["bucket_limit"])
    counts = np.bincount(np.searchsorted(edges, vals, side="right"), minlength=len(edges))
    assert counts.tolist() == [int(c) for c in h["bucket"]]

    with tf.Graph().as_default():
        tf.set_random_seed(1)
        w = tf.Variable(tf.random_normal([30, 10]), name="W")
        tf.summary.histogram("weights", w)
        tf.summary.scalar("mean", tf.reduce_mean(w))
        merged = tf.summary.merge_all()
        writer = tf.summary.FileWriter(str(tmp_path))
        with tf.Session(config=tf.ConfigProto(device="cpu", device_filters=["/job:ps"],
                                              gpu_options=tf.GPUOptions(allow_growth=True))) as sess:
            sess.run(tf.global_variables_initializer())
            data, wv = sess.run([merged, w])
            writer.add_summary(data, 3)
        writer.close()
    from distributed_tensorflow_amd.summary import read_records
    from distributed_tensorflow_amd.utils import protowire as pw
    path = [os.path.join(str(tmp_path), f) for f in os.listdir(tmp_path)][0]
    summaries = [pw.parse(r)[5][0] for r in read_records(path) if 5 in pw.parse(r)]
    (tag, h), = decode_histograms(summaries[0])
    assert tag == "weights" and h["num"] == 300 and np.isclose(h["min"], wv.min())
