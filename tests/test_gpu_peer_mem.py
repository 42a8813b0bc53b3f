"""Peer-memory collectives on real hardware: four processes sharing GPU 0 map each other's
buffers through HIP IPC and run the direct all-gather / all-reduce and the sharded GEMM built on
them (same-device peers: the one-GPU box cannot host two RCCL ranks)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "peer_worker.py")


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn(world, tmp_path, **extra_env):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT,
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **extra_env)
        procs.append(subprocess.Popen([sys.executable, "-u", WORKER, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_dead_peer_times_out_into_comm_error(tmp_path):
    """Device-side completion is bounded: a peer that stalls and never pushes leaves rank 0's
    wait kernel to give up after the configured time, and the next check raises CommError."""
    procs, outs = _spawn(2, tmp_path, DTF_PEER_TEST="dead_peer", DTF_PEER_TIMEOUT_S="2")
    assert procs[0].returncode == 0, outs[0][-3000:]
    with open(tmp_path / "rank0.json") as fh:
        res = json.load(fh)
    assert res["sync"] == "flags"
    assert res["raised"] is True and "did not signal" in res["message"]
    assert 1.5 < res["seconds"] < 30.0


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("world,sync", [(2, "flags"), (4, "flags"), (2, "host")])
def test_peer_memory_collectives_same_device(world, sync, tmp_path):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT,
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", DTF_PEER_SYNC=sync)
        procs.append(subprocess.Popen([sys.executable, "-u", WORKER, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, outs[rank][-3000:])
    for rank in range(world):
        with open(tmp_path / ("rank%d.json" % rank)) as fh:
            res = json.load(fh)
        assert res["world"] == world
        assert res["sync"] == sync
        if sync == "flags":
            # completion is device-side: no host stream synchronisation in any exchange
            assert res["host_syncs_all_gather"] == 0 and res["host_syncs_pipelined"] == 0
        assert res["all_gather_ok"], res
        assert res["all_reduce_ok"], res
        scale = res["ref_scale"]
        # the all-gather variants run the same kernel with the same K order on row blocks of
        # the single-rank product: identical bits
        assert res["gemm_errs"]["m_shard_allgather_direct"] == 0.0, res
        assert res["gemm_errs"]["pipelined_direct"] == 0.0, res
        # the K split sums bf16 partial products of K/world each: another rounding, bounded
        assert res["gemm_errs"]["k_shard_allreduce_direct"] <= 0.02 * scale, res
        # recomputed heads come from the same kernel with the same K order: identical bits
        assert res["gemm_errs"]["recompute_direct"] == 0.0, res
        assert res["gemm_errs"]["recompute_rccl_layout"] == 0.0, res
        assert res["gemm_errs"]["kernel_transport"] == 0.0, res
        # rows stored into the peers by the product's own epilogue: identical bits again, and
        # under device-side completion no host synchronisation either
        assert res["gemm_errs"]["epilogue_transport"] == 0.0, res
        assert res["gemm_errs"]["epilogue_transport_pipelined"] == 0.0, res
        if sync == "flags":
            assert res["epilogue_transport_host_syncs"] == 0
            assert res["epilogue_transport_pipelined_host_syncs"] == 0


FACADE_WORKER = os.path.join(ROOT, "tests", "helpers", "facade_shard_worker.py")


def _spawn_facade(world, tmp_path, **extra_env):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT,
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **extra_env)
        procs.append(subprocess.Popen([sys.executable, "-u", FACADE_WORKER, str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=300)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, outs[rank][-3000:])
    results = []
    for rank in range(world):
        with open(tmp_path / ("rank%d.json" % rank)) as fh:
            results.append(json.load(fh))
    return results


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("world,dtype,mode,collective", [
    (2, "fp32", "m", "peer"), (4, "bf16", "m", "peer"), (2, "bf16", "m", "rccl"),
    (4, "fp32", "2d", "rccl"), (2, "bf16", "k", "rccl"), (2, "fp8", "m", "peer")])
def test_sharded_tf_matmul_rehearsed_on_one_gpu(world, dtype, mode, collective, tmp_path):
    """``tf.matmul`` of two variables, planned as a sharded product and run by the benchmark's
    engine, with 2 and 4 ranks sharing GPU 0: row-major C on every rank equal to the product one
    rank computes alone (bit for bit except the K split), the variables stored as blocks, no
    torch kernel, and -- over peer memory -- the row panels' transfers running while the next
    panel is multiplied (device brackets of a traced step)."""
    # (the fp32 case asserts overlap below: panels of 2048 x 4096 x 4096, ~0.7 ms each, long
    # against the time the host needs to queue the copies behind them)
    m, n, k = (8192, 4096, 4096) if dtype == "fp32" and collective == "peer" else (4096, 2048, 2048)
    overlap_case = dtype == "fp32" and collective == "peer"
    panels = 4 if overlap_case else 2          # (three hand-overs in which a push can overlap)
    results = _spawn_facade(world, tmp_path, DTF_SHAPE="%d,%d,%d" % (m, n, k), DTF_DTYPE=dtype,
                            DTF_SHARD=mode, DTF_COLLECTIVE=collective, DTF_PANELS=str(panels))
    want = {("m", "peer"): "m_shard_allgather_direct", ("m", "rccl"): "m_shard_allgather",
            ("2d", "rccl"): "block_2d_allgather", ("k", "rccl"): "k_shard_allreduce"}
    for res in results:
        assert res["world"] == world and res["strategy"] == want[(mode, collective)], res
        assert res["torch_fallbacks"] == 0, res
        if mode == "k":
            assert res["max_err"] <= 0.02 * res["scale"], res
        else:
            assert res["bit_equal"], res
        if mode == "m":
            assert res["a_local"] == [m // world, k]
        assert res["panel_products"] >= (2 if mode != "k" or world else 1)
        if collective == "peer":
            assert res["panels"] == panels and res["transfers"] >= panels * (world - 1)
            # panel 0's pushes are queued behind its product and run under panel 1's. Asserted
            # where a panel's product is long against the host's time to queue the copies; the
            # short bf16 / fp8 panels of the small shape are over before an eagerly issued,
            # traced step gets the next launch out
    if overlap_case:
        # (the ranks time-slice ONE GPU here and their copies are kernels that need free CUs: a
        # single rank can come out with its pushes serialised behind its products, so the
        # overlap is asked of the run, not of every rank)
        assert any(res["overlap_us"] > 0.0 for res in results), results
