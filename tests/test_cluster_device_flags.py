"""T0: ClusterSpec, device strings, replica_device_setter, flags."""
import json
import sys

import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.app import flags
from distributed_tensorflow_amd.framework.device import DeviceSpec, merge_device
from distributed_tensorflow_amd.train.cluster import ClusterDef, ClusterSpec, task_from_env

PS = ["10.0.1.2:2222"]
W5 = ["10.0.1.%d:2222" % i for i in range(3, 8)]


def test_cluster_spec_reference_shape():
    c = tf.train.ClusterSpec({"ps": PS, "worker": W5})
    assert c.jobs == ["ps", "worker"]
    assert c.num_tasks("worker") == 5 and c.num_tasks("ps") == 1
    assert c.task_indices("worker") == [0, 1, 2, 3, 4]
    assert c.task_address("worker", 3) == "10.0.1.6:2222"
    assert c.job_tasks("ps") == PS
    assert c.as_dict() == {"ps": PS, "worker": W5}
    assert bool(c) and not bool(ClusterSpec({}))
    assert c == ClusterSpec(c) and c == ClusterSpec(c.as_cluster_def())
    assert c != ClusterSpec({"ps": PS, "worker": W5[:4]})
    assert "worker" in repr(c)


def test_cluster_spec_sparse_and_errors():
    c = ClusterSpec({"worker": {1: "a:1", 3: "b:2"}})
    assert c.task_indices("worker") == [1, 3]
    assert c.job_tasks("worker") == [None, "a:1", None, "b:2"]
    assert c.as_dict() == {"worker": {1: "a:1", 3: "b:2"}}
    with pytest.raises(ValueError):
        c.task_address("worker", 0)
    with pytest.raises(ValueError):
        c.num_tasks("ps")
    with pytest.raises(TypeError):
        ClusterSpec({"worker": "not-a-list"})
    with pytest.raises(TypeError):
        ClusterSpec([1, 2])
    with pytest.raises(TypeError):
        ClusterSpec({"worker": [5]})


def test_cluster_def_wire_roundtrip():
    c = ClusterSpec({"ps": PS, "worker": {0: "h:1", 2: "h:3"}})
    data = c.as_cluster_def().SerializeToString()
    assert ClusterSpec(ClusterDef.FromString(data)) == c


def test_cluster_from_cli_and_env():
    c = ClusterSpec.from_host_lists("p:1", "w0:1, w1:2,w2:3")
    assert c.as_dict() == {"ps": ["p:1"], "worker": ["w0:1", "w1:2", "w2:3"]}
    env = {"TF_CONFIG": json.dumps({"cluster": {"worker": ["a:1", "b:2"]},
                                    "task": {"type": "worker", "index": 1}})}
    assert ClusterSpec.from_tf_config(env).num_tasks("worker") == 2
    assert task_from_env(env) == ("worker", 1)
    env = {"WORLD_SIZE": "4", "RANK": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500"}
    c = ClusterSpec.from_launcher_env(env, num_ps=2)
    assert c.num_tasks("worker") == 4 and c.num_tasks("ps") == 2
    assert task_from_env(env) == ("worker", 2)
    assert c.global_rank("worker", 3) == 3 and c.global_rank("ps", 0) == 4


def test_device_spec_parse_format_merge():
    d = DeviceSpec.from_string("/job:worker/replica:0/task:3/device:GPU:1")
    assert (d.job, d.replica, d.task, d.device_type, d.device_index) == ("worker", 0, 3, "GPU", 1)
    assert d.to_string() == "/job:worker/replica:0/task:3/device:GPU:1"
    assert DeviceSpec.from_string("/gpu:0").to_string() == "/device:GPU:0"
    assert DeviceSpec.from_string("/job:ps").to_string() == "/job:ps"
    assert merge_device("/job:worker/task:1", "/device:GPU:0") == "/job:worker/task:1/device:GPU:0"
    assert merge_device("/job:worker/task:1", "/job:ps") == "/job:ps/task:1"
    assert DeviceSpec.from_string("/job:ps").is_subset_of(DeviceSpec.from_string("/job:ps/task:2"))
    for bad in ("job:ps", "/job:ps/task:x", "/bogus:1", "/task:-1"):
        with pytest.raises(ValueError):
            DeviceSpec.from_string(bad)


_jobs = st.sampled_from([None, "ps", "worker"])
_idx = st.one_of(st.none(), st.integers(0, 7))
_spec = st.builds(DeviceSpec, job=_jobs, replica=_idx, task=_idx,
                  device_type=st.sampled_from([None, "CPU", "GPU"]), device_index=_idx)


@settings(max_examples=200, deadline=None)
@given(_spec, _spec)
def test_device_merge_properties(a, b):
    # string round trip (an index without a type is not representable: normalise first)
    for s in (a, b):
        if s.device_type is None:
            s.device_index = None
    assert DeviceSpec.from_string(a.to_string()) == a
    m = a.merge_from(b)
    for f in ("job", "replica", "task", "device_type"):
        assert getattr(m, f) == (getattr(b, f) if getattr(b, f) is not None else getattr(a, f))
    assert a.merge_from(DeviceSpec()) == a            # identity
    assert m.merge_from(b) == m                       # idempotent


def test_replica_device_setter_placement():
    cluster = tf.train.ClusterSpec({"ps": ["p0:1", "p1:1"], "worker": ["w0:1", "w1:1"]})
    with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:1",
                                                  cluster=cluster)):
        a = tf.Variable(tf.zeros([3]))
        b = tf.Variable(tf.zeros([3]))
        c = tf.Variable(tf.zeros([3]))
        x = tf.placeholder(tf.float32, [None, 3])
        y = x + a
        upd = a.assign_add(tf.ones([3]))
    assert a.device == "/job:ps/task:0" and b.device == "/job:ps/task:1"
    assert c.device == "/job:ps/task:0"                      # round-robin wraps
    assert x.device == "/job:worker/task:1" and y.device == "/job:worker/task:1"
    assert a.initializer.device == a.device                  # Assign follows the variable
    assert upd.op.device == a.device                         # so the update runs on the owner
    assert a.initial_value.device == "/job:worker/task:1"    # the value is computed on the worker


def test_replica_device_setter_options():
    assert tf.train.replica_device_setter(ps_tasks=0) is None
    assert tf.train.replica_device_setter(cluster={"worker": ["w:1"]}) is None
    fn = tf.train.replica_device_setter(ps_tasks=3, ps_ops=["Placeholder"])
    with tf.device(fn):
        v = tf.Variable(tf.zeros([2]))
        p = tf.placeholder(tf.float32, [2])
    assert v.device == "/job:worker" and p.device == "/job:ps/task:0"
    greedy = tf.train.GreedyLoadBalancingStrategy(2)
    with tf.device(tf.train.replica_device_setter(ps_tasks=2, ps_strategy=greedy)):
        big = tf.Variable(tf.zeros([784, 100]))
        s1 = tf.Variable(tf.zeros([10]))
        s2 = tf.Variable(tf.zeros([100]))
    assert big.device == "/job:ps/task:0"
    assert s1.device == "/job:ps/task:1" and s2.device == "/job:ps/task:1"
    with tf.device(tf.train.replica_device_setter(ps_tasks=1)):
        with tf.device("/device:GPU:0"):
            q = tf.placeholder(tf.float32, [1])
        with tf.device(None):
            r = tf.placeholder(tf.float32, [1])
    assert q.device == "/job:worker/device:GPU:0" and r.device == ""


def test_flags_parsing():
    flags.DEFINE_string("job_name", "ps", "role")
    flags.DEFINE_integer("task_index", 0, "index")
    flags.DEFINE_float("lr", 0.5, "")
    flags.DEFINE_boolean("verbose", False, "")
    F = flags.FLAGS
    rest = F._parse(["--job_name=worker", "--task_index", "3", "--lr=0.25", "--verbose",
                     "positional", "--unknown=1"])
    assert (F.job_name, F.task_index, F.lr, F.verbose) == ("worker", 3, 0.25, True)
    assert rest == ["positional", "--unknown=1"]
    F._parse(["--noverbose"])
    assert F.verbose is False and F.job_name == "ps"       # unspecified flags fall back to defaults
    F.task_index = 7                                        # explicit assignment wins over argv
    F._parse(["--task_index=1"])
    assert F.task_index == 7
    with pytest.raises(AttributeError):
        F.nope
    with pytest.raises(ValueError):
        F._parse(["--task_index"])
    assert "job_name" in F and "--lr" in F.get_help()


def test_flags_lazy_parse_and_app_run(monkeypatch):
    monkeypatch.setattr("sys.argv", ["prog", "--job_name=worker", "--task_index=2", "extra"])
    flags.define_cluster_flags("ps")
    assert flags.FLAGS.job_name == "worker" and flags.FLAGS.task_index == 2
    seen = {}

    def main(argv):
        seen["argv"] = argv
        return 3

    with pytest.raises(SystemExit) as e:
        tf.app.run(main=main)
    assert e.value.code == 3 and seen["argv"] == ["prog", "extra"]
    with pytest.raises(SystemExit):
        tf.app.run(main=main, argv=["p", "--ps_hosts=a:1,b:2"])
    assert flags.FLAGS.ps_hosts == "a:1,b:2"


def test_flags_help_prints_and_exits(monkeypatch, capsys):
    """--help on the process's command line lists the declared flags and exits when the flags
    are first read (so an example started with --help never reaches ``server.join()``)."""
    from distributed_tensorflow_amd.app import flags as F
    F.FLAGS._reset()
    try:
        F.DEFINE_string("job_name", "ps", "Either 'ps' or 'worker'")
        monkeypatch.setattr(sys, "argv", ["prog", "--help"])
        with pytest.raises(SystemExit) as e:
            F.FLAGS.job_name
        assert e.value.code == 0
        assert "--job_name: Either 'ps' or 'worker' (default: 'ps')" in capsys.readouterr().out
        # an explicit argv (a library caller's list) is not the command line: left alone
        assert F.FLAGS._parse(["--help", "x"]) == ["--help", "x"]
    finally:
        F.FLAGS._reset()
