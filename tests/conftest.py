import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-process / long-running CPU test")


@pytest.fixture(autouse=True)
def _fresh_graph():
    """Every test builds into its own default graph and a clean flag registry."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.app import flags
    g = tf.Graph()
    with g.as_default():
        yield g
    flags.FLAGS._reset()
