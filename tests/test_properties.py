"""Property tests (hypothesis): device-spec algebra, shard arithmetic, wire formats."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from distributed_tensorflow_amd.framework.device import DeviceSpec, merge_device
from distributed_tensorflow_amd.parallel.sharded_matmul import _split
from distributed_tensorflow_amd.utils import crc32c, protowire as pw
from distributed_tensorflow_amd.utils import sstable

jobs = st.sampled_from([None, "ps", "worker", "chief"])
small = st.one_of(st.none(), st.integers(0, 7))
dtypes_ = st.sampled_from([None, "CPU", "GPU"])


@st.composite
def specs(draw):
    dt = draw(dtypes_)
    return DeviceSpec(job=draw(jobs), replica=draw(small), task=draw(small), device_type=dt,
                      device_index=draw(small) if dt else None)


@given(specs())
def test_device_spec_string_round_trip(spec):
    assert DeviceSpec.from_string(spec.to_string()) == spec


@given(specs(), specs())
def test_device_merge_inner_fields_win(outer, inner):
    merged = DeviceSpec.from_string(merge_device(outer.to_string(), inner.to_string()))
    for field in ("job", "replica", "task", "device_type"):
        want = getattr(inner, field) if getattr(inner, field) is not None else getattr(outer, field)
        assert getattr(merged, field) == want
    # merging with the empty spec is the identity, on either side
    assert merge_device(outer.to_string(), "") == outer.to_string()
    assert merge_device("", inner.to_string()) == inner.to_string()


@given(specs(), specs(), specs())
def test_device_merge_is_associative(a, b, c):
    s = [x.to_string() for x in (a, b, c)]
    assert merge_device(merge_device(s[0], s[1]), s[2]) == merge_device(s[0], merge_device(s[1], s[2]))


@given(st.integers(1, 10000), st.integers(1, 64))
def test_split_partitions_exactly(total, parts):
    blocks = [_split(total, parts, i) for i in range(parts)]
    assert blocks[0][0] == 0 and sum(sz for _, sz in blocks) == total
    for (s0, n0), (s1, _n1) in zip(blocks, blocks[1:]):
        assert s0 + n0 == s1                      # contiguous, in order
    sizes = [n for _, n in blocks]
    assert max(sizes) - min(sizes) <= 1           # near-equal


@given(st.integers(0, 2 ** 64 - 1))
def test_varint_round_trip(v):
    data = pw.encode_varint(v)
    got, pos = pw.decode_varint(data, 0)
    assert got == v and pos == len(data) and len(data) <= 10


@given(st.binary(max_size=300), st.binary(max_size=300))
def test_crc32c_is_incremental_and_mask_is_invertible(a, b):
    assert crc32c.crc32c(a + b) == crc32c.crc32c(b, crc32c.crc32c(a))
    c = crc32c.crc32c(a)
    assert crc32c.unmask(crc32c.mask(c)) == c


@settings(max_examples=40, deadline=None)
@given(st.dictionaries(st.binary(min_size=1, max_size=40), st.binary(max_size=200), max_size=60))
def test_sstable_round_trip(tmp_path_factory, table):
    path = str(tmp_path_factory.mktemp("sst") / "t.index")
    rows = sorted(table.items())
    sstable.write_table(path, rows)
    assert sstable.read_table(path) == rows
