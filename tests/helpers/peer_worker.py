"""Rank body for the peer-memory collective tests: N processes that all use GPU 0.

RCCL refuses two ranks on one device, so the control plane here is gloo; the data plane -- HIP
IPC mapped buffers, peer copies, the reduce kernel -- is the code that runs between GPUs of a
node. Each rank writes ``<out_dir>/rank<r>.json``."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.parallel.comm import SingleComm, TorchDistComm  # noqa: E402
from distributed_tensorflow_amd.parallel.peer_mem import (SymmetricBuffer, all_gather_rows,  # noqa: E402
                                                          all_reduce_rows)
from distributed_tensorflow_amd.parallel.sharded_matmul import ShardedMatmul  # noqa: E402


def main() -> int:
    out_dir = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = TorchDistComm(device=dev, backend="gloo")
    rank, world = comm.rank, comm.world_size
    res = {"rank": rank, "world": world}

    if os.environ.get("DTF_PEER_TEST") == "dead_peer":
        # rank 1 stalls and never pushes: rank 0's wait kernel must give up after
        # DTF_PEER_TIMEOUT_S and the next check must raise CommError (no hang, no host barrier)
        from distributed_tensorflow_amd.parallel.comm import CommError
        buf = SymmetricBuffer(comm, (4 * world, 256), torch.float32, dev)
        res["sync"] = buf.sync
        import time
        if rank != 0:
            with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
                json.dump(res, fh)
            time.sleep(8.0)            # alive (its memory stays mapped) but never pushes
            os._exit(0)
        t0 = time.time()
        buf.push_rows(0, 4)
        buf.finish()
        torch.cuda.synchronize()
        try:
            buf.check()
            res["raised"] = False
        except CommError as e:
            res["raised"] = True
            res["message"] = str(e)
        res["seconds"] = time.time() - t0
        with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
            json.dump(res, fh)
        os._exit(0)

    # all-gather: rank r owns rows [4r, 4r+4)
    buf = SymmetricBuffer(comm, (4 * world, 256), torch.float32, dev)
    buf.local.zero_()
    buf.quiesce()                                  # nobody pushes into a buffer being zeroed
    buf.local[4 * rank:4 * rank + 4] = float(rank + 1)
    for _ in range(3):                             # repeated use of the same buffer
        all_gather_rows(buf, 4)
    want = torch.arange(1, world + 1, dtype=torch.float32, device=dev).repeat_interleave(4)
    res["all_gather_ok"] = bool((buf.local == want[:, None]).all().item())
    res["sync"] = buf.sync
    res["host_syncs_all_gather"] = buf.host_syncs

    # two-shot all-reduce of bf16 partials
    part = SymmetricBuffer(comm, (8 * world, 128), torch.bfloat16, dev)
    total = SymmetricBuffer(comm, (8 * world, 128), torch.bfloat16, dev)
    rows = torch.arange(8 * world, device=dev, dtype=torch.float32)[:, None]
    part.local.copy_(((rank + 1) * (rows + 1)).expand(-1, 128))
    all_reduce_rows(part, total)
    s = world * (world + 1) / 2
    want = (s * (rows + 1)).expand(-1, 128).to(torch.bfloat16)
    res["all_reduce_ok"] = bool((total.local == want).all().item())

    # sharded GEMM over peer memory against the single-process product
    m, n, k = 1024, 512, 256
    ref = ShardedMatmul(m, n, k, SingleComm(dev), dtype=torch.bfloat16, strategy="replicated",
                        seed=5, device=dev)
    ref.init_operands()
    want = ref.step().float()
    errs = {}
    for strategy, panels in (("m_shard_allgather_direct", 2), ("k_shard_allreduce_direct", None)):
        eng = ShardedMatmul(m, n, k, comm, dtype=torch.bfloat16, strategy=strategy, seed=5,
                            device=dev, panels=panels)
        eng.init_operands()
        for _ in range(2):
            got = eng.step()
        torch.cuda.synchronize()
        errs[strategy] = (got.float() - want).abs().max().item()
    eng = ShardedMatmul(m, n, k, comm, dtype=torch.bfloat16, strategy="m_shard_allgather_direct",
                        seed=5, device=dev, panels=2, pipelined=True)
    eng.init_operands()
    for _ in range(5):
        eng.step()
    eng.wait()
    torch.cuda.synchronize()
    errs["pipelined_direct"] = (eng.result_dense().float() - want).abs().max().item()
    res["host_syncs_pipelined"] = sum(sym.host_syncs for sym in eng._sym_sets)
    # recomputed heads: only rows x.. of every block travel; every rank computes the heads itself
    for name, strategy in (("recompute_direct", "m_shard_allgather_direct"),
                           ("recompute_rccl_layout", "m_shard_allgather")):
        eng = ShardedMatmul(m, n, k, comm, dtype=torch.bfloat16, strategy=strategy, seed=5,
                            device=dev, panels=1, pipelined=True, recompute_rows=128)
        eng.init_operands()
        for _ in range(3):
            eng.step()
        eng.wait()
        torch.cuda.synchronize()
        errs[name] = (eng.result_dense().float() - want).abs().max().item()
    # the kernel transport: one peer_push launch stores the rows into every peer
    eng = ShardedMatmul(m, n, k, comm, dtype=torch.bfloat16, strategy="m_shard_allgather_direct",
                        seed=5, device=dev, panels=2, pipelined=True, recompute_rows=128,
                        peer_transport="kernel")
    eng.init_operands()
    for _ in range(3):
        eng.step()
    eng.wait()
    torch.cuda.synchronize()
    errs["kernel_transport"] = (eng.result_dense().float() - want).abs().max().item()
    # the epilogue transport: no transfer step, the product's own epilogue stores every result
    # vector into all peers; plain and pipelined with recomputed heads (which stay local)
    for name, kw in (("epilogue_transport", dict(panels=2)),
                     ("epilogue_transport_pipelined",
                      dict(panels=2, pipelined=True, recompute_rows=128))):
        eng = ShardedMatmul(m, n, k, comm, dtype=torch.bfloat16,
                            strategy="m_shard_allgather_direct", seed=5, device=dev,
                            peer_transport="epilogue", **kw)
        eng.init_operands()
        for _ in range(3):
            eng.step()
        eng.wait()
        torch.cuda.synchronize()
        errs[name] = (eng.result_dense().float() - want).abs().max().item()
        res[name + "_host_syncs"] = sum(sym.host_syncs for sym in eng._sym_sets)
    res["gemm_errs"] = errs
    res["ref_scale"] = want.abs().max().item()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
        json.dump(res, fh)
    comm.barrier()
    comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
