"""MNIST input pipeline: ``read_data_sets(dir, one_hot=True)`` -> ``.train/.validation/.test``.

The reference loads MNIST in every process through the TF tutorial helper
(``tf_distributed.py:27-28``): 55 000 train / 5 000 validation / 10 000 test examples, images
``[N,784] float32`` in [0,1], labels one-hot, ``next_batch(n)`` walking a per-epoch shuffle
(``tf_distributed.py:108``). This module reads the same IDX files when they are present in
``train_dir`` (``train-images-idx3-ubyte[.gz]`` etc.; there is no downloading -- GPU machines
have no network) and otherwise generates a *synthetic* MNIST-shaped dataset: ten fixed class
prototypes plus noise, so a linear/MLP model trained on it shows a falling loss and a rising
accuracy just as on the real data. ``Datasets.synthetic`` tells which one you got.
"""
from __future__ import annotations

import gzip
import os
import struct
from typing import NamedTuple, Optional, Tuple

import numpy as np

TRAIN_SIZE, VALIDATION_SIZE, TEST_SIZE = 55000, 5000, 10000
IMAGE_PIXELS, NUM_CLASSES = 784, 10


class DataSet:
    def __init__(self, images: np.ndarray, labels: np.ndarray, seed: Optional[int] = None):
        assert images.shape[0] == labels.shape[0]
        self._images = images
        self._labels = labels
        self._num_examples = images.shape[0]
        self._epochs_completed = 0
        self._index_in_epoch = 0
        self._rng = np.random.RandomState(seed)
        self._perm = np.arange(self._num_examples)
        self._dev_images = None      # set by to_device(): the whole set resident in HBM
        self._dev_labels = None
        self._dev_perm = None

    def to_device(self, device) -> "DataSet":
        """Keep the whole set in GPU memory (MNIST train is 172 MB of fp32): ``next_batch`` then
        gathers the batch on the device and returns torch tensors, so a step's feed is a
        device-to-device copy instead of two host-to-device transfers; ``images`` / ``labels``
        handed to ``feed_dict`` likewise stay on the device. The shuffle order is the host
        permutation, so batches are identical to the host path's."""
        import torch
        dev = torch.device(device)
        self._dev_images = torch.from_numpy(np.ascontiguousarray(self._images)).to(dev)
        self._dev_labels = torch.from_numpy(np.ascontiguousarray(self._labels)).to(dev)
        self._dev_perm = None
        return self

    @property
    def device_images(self):
        return self._dev_images

    @property
    def device_labels(self):
        return self._dev_labels

    @property
    def images(self) -> np.ndarray:
        return self._images

    @property
    def labels(self) -> np.ndarray:
        return self._labels

    @property
    def num_examples(self) -> int:
        return self._num_examples

    @property
    def epochs_completed(self) -> int:
        return self._epochs_completed

    def next_batch(self, batch_size: int, shuffle: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """Next ``batch_size`` examples; reshuffles at every epoch boundary (a batch that
        straddles the boundary takes the rest of the old epoch plus the head of the new)."""
        if batch_size > self._num_examples:
            raise ValueError("batch_size %d exceeds dataset size %d"
                             % (batch_size, self._num_examples))
        start = self._index_in_epoch
        if self._epochs_completed == 0 and start == 0 and shuffle:
            self._perm = self._rng.permutation(self._num_examples)
        if start + batch_size > self._num_examples:
            self._epochs_completed += 1
            rest = self._perm[start:]
            if shuffle:
                self._perm = self._rng.permutation(self._num_examples)
            take = batch_size - rest.shape[0]
            self._index_in_epoch = take
            idx = np.concatenate([rest, self._perm[:take]])
        else:
            self._index_in_epoch = start + batch_size
            idx = self._perm[start:start + batch_size]
        if self._dev_images is not None:
            import torch
            if self._dev_perm is None or self._dev_perm[0] is not self._perm:
                # once per epoch: the whole set gathered into the epoch's order on the device
                # (one permutation upload + two gathers, ~0.1 ms for MNIST train), after which a
                # batch is a contiguous slice -- a view, no kernel launch per step
                dperm = torch.from_numpy(self._perm).to(self._dev_images.device)
                self._dev_perm = (self._perm, self._dev_images.index_select(0, dperm),
                                  self._dev_labels.index_select(0, dperm))
            if idx.base is self._perm or idx is self._perm:
                return (self._dev_perm[1][start:start + batch_size],
                        self._dev_perm[2][start:start + batch_size])
            # the batch that straddles an epoch boundary: gathered from the resident set
            didx = torch.from_numpy(idx).to(self._dev_images.device)
            return (self._dev_images.index_select(0, didx),
                    self._dev_labels.index_select(0, didx))
        return self._images[idx], self._labels[idx]


class Datasets(NamedTuple):
    train: DataSet
    validation: DataSet
    test: DataSet
    synthetic: bool


def _open(path: str):
    return gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")


def _find(train_dir: str, stem: str) -> Optional[str]:
    for name in (stem, stem + ".gz", stem.replace("-idx", ".idx"), stem.replace("-idx", ".idx") + ".gz"):
        path = os.path.join(train_dir, name)
        if os.path.exists(path):
            return path
    return None


def read_idx_images(path: str) -> np.ndarray:
    with _open(path) as fh:
        magic, n, rows, cols = struct.unpack(">IIII", fh.read(16))
        if magic != 2051:
            raise ValueError("bad magic %d in MNIST image file %s" % (magic, path))
        data = np.frombuffer(fh.read(n * rows * cols), dtype=np.uint8)
    return (data.reshape(n, rows * cols).astype(np.float32) / 255.0)


def read_idx_labels(path: str) -> np.ndarray:
    with _open(path) as fh:
        magic, n = struct.unpack(">II", fh.read(8))
        if magic != 2049:
            raise ValueError("bad magic %d in MNIST label file %s" % (magic, path))
        return np.frombuffer(fh.read(n), dtype=np.uint8).astype(np.int64)


def dense_to_one_hot(labels: np.ndarray, num_classes: int = NUM_CLASSES) -> np.ndarray:
    out = np.zeros((labels.shape[0], num_classes), dtype=np.float64)
    out[np.arange(labels.shape[0]), labels] = 1.0
    return out


def synthetic_mnist(n: int, seed: int = 0, noise: float = 0.35) -> Tuple[np.ndarray, np.ndarray]:
    """``n`` MNIST-shaped examples: class prototypes (sparse strokes) + uniform noise, in [0,1]."""
    proto_rng = np.random.RandomState(1234)          # the prototypes are the same for every split
    protos = (proto_rng.rand(NUM_CLASSES, IMAGE_PIXELS) < 0.18).astype(np.float32) \
        * (0.6 + 0.4 * proto_rng.rand(NUM_CLASSES, IMAGE_PIXELS).astype(np.float32))
    rng = np.random.RandomState(seed)
    labels = rng.randint(0, NUM_CLASSES, size=n).astype(np.int64)
    images = protos[labels] + noise * (rng.rand(n, IMAGE_PIXELS).astype(np.float32) - 0.5)
    return np.clip(images, 0.0, 1.0).astype(np.float32), labels


def read_data_sets(train_dir: str = "MNIST_data", one_hot: bool = False,
                   validation_size: int = VALIDATION_SIZE, seed: Optional[int] = None,
                   synthetic: Optional[bool] = None, train_size: int = TRAIN_SIZE + VALIDATION_SIZE,
                   test_size: int = TEST_SIZE) -> Datasets:
    files = {k: _find(train_dir, stem) for k, stem in (
        ("train_images", "train-images-idx3-ubyte"), ("train_labels", "train-labels-idx1-ubyte"),
        ("test_images", "t10k-images-idx3-ubyte"), ("test_labels", "t10k-labels-idx1-ubyte"))}
    have_files = all(files.values())
    if synthetic is None:
        synthetic = not have_files
    if synthetic:
        train_images, train_labels = synthetic_mnist(train_size, seed=1)
        test_images, test_labels = synthetic_mnist(test_size, seed=2)
    else:
        if not have_files:
            raise FileNotFoundError("MNIST IDX files not found in %r" % train_dir)
        train_images = read_idx_images(files["train_images"])
        train_labels = read_idx_labels(files["train_labels"])
        test_images = read_idx_images(files["test_images"])
        test_labels = read_idx_labels(files["test_labels"])
    if not 0 <= validation_size <= train_images.shape[0]:
        raise ValueError("validation_size must be between 0 and %d" % train_images.shape[0])
    if one_hot:
        train_labels, test_labels = dense_to_one_hot(train_labels), dense_to_one_hot(test_labels)
    val_images, val_labels = train_images[:validation_size], train_labels[:validation_size]
    train_images, train_labels = train_images[validation_size:], train_labels[validation_size:]
    return Datasets(train=DataSet(train_images, train_labels, seed),
                    validation=DataSet(val_images, val_labels, seed),
                    test=DataSet(test_images, test_labels, seed), synthetic=bool(synthetic))
