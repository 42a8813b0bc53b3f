"""Single-node launcher: start every task of a cluster as a child process.

The reference is started by hand, one ``python tf_distributed.py --job_name=... --task_index=...``
per machine (``tf_distributed.py:4-6``). On one MI355X node the natural layout is one worker
process per GPU, so this launcher synthesises the cluster on localhost ports and spawns::

    python -m distributed_tensorflow_amd.launch --num_workers 8 --protocol rccl \
        examples/tf_distributed.py -- --training_epochs 1

* ``--protocol grpc`` (default): ``--num_ps`` parameter-server tasks + ``--num_workers`` workers;
  every task receives ``--ps_hosts/--worker_hosts/--job_name/--task_index`` and workers get
  ``--shutdown_ps`` so the ps tasks leave ``join()`` when all workers are done.
* ``--protocol rccl``: workers only, one per GPU; each child gets ``RANK``, ``LOCAL_RANK``,
  ``WORLD_SIZE``, ``MASTER_ADDR=127.0.0.1``, ``MASTER_PORT`` so the script's
  ``ClusterSpec.from_launcher_env()`` / ``torch.distributed`` rendezvous work unchanged.

Children are ordinary subprocesses (never an exec of this process); their output is prefixed
with ``[job:index]``. The launcher exits with the first non-zero child status, terminating the
remaining children by pid.
"""
from __future__ import annotations

import argparse
import os
import signal
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple


def _free_ports(n: int) -> List[int]:
    import socket
    socks, ports = [], []
    for _ in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        socks.append(s)
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    return ports


def plan_tasks(num_workers: int, num_ps: int, protocol: str,
               ports: Optional[Sequence[int]] = None) -> Tuple[Dict[str, List[str]], List[Tuple[str, int]]]:
    """Return ``(cluster_dict, [(job, index), ...])`` for a localhost cluster."""
    if num_workers < 1:
        raise ValueError("need at least one worker")
    if protocol == "rccl":
        num_ps = 0
    need = num_workers + num_ps
    ports = list(ports) if ports is not None else _free_ports(need)
    if len(ports) < need:
        raise ValueError("need %d ports, got %d" % (need, len(ports)))
    cluster: Dict[str, List[str]] = {}
    if num_ps:
        cluster["ps"] = ["127.0.0.1:%d" % p for p in ports[:num_ps]]
    cluster["worker"] = ["127.0.0.1:%d" % p for p in ports[num_ps:need]]
    tasks = [("ps", i) for i in range(num_ps)] + [("worker", i) for i in range(num_workers)]
    return cluster, tasks


def child_command(script: str, script_args: Sequence[str], cluster: Dict[str, List[str]],
                  job: str, index: int, protocol: str, strategy: Optional[str]) -> List[str]:
    cmd = [sys.executable, "-u", script,
           "--job_name=%s" % job, "--task_index=%d" % index,
           "--worker_hosts=%s" % ",".join(cluster["worker"]),
           "--protocol=%s" % protocol]
    if cluster.get("ps"):
        cmd.append("--ps_hosts=%s" % ",".join(cluster["ps"]))
        if job == "worker":
            cmd.append("--shutdown_ps")
    if strategy:
        cmd.append("--strategy=%s" % strategy)
    return cmd + list(script_args)


def child_env(cluster: Dict[str, List[str]], job: str, index: int, protocol: str) -> Dict[str, str]:
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # every task of the job runs on this host: split the cores between them
    ntasks = sum(len(v) for v in cluster.values())
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, ntasks))))
    if protocol == "rccl" and job == "worker":
        chief_port = cluster["worker"][0].rpartition(":")[2]
        env.update(RANK=str(index), LOCAL_RANK=str(index),
                   WORLD_SIZE=str(len(cluster["worker"])),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=chief_port)
    else:
        for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(key, None)
    return env


def _pump(proc: subprocess.Popen, tag: str, sink) -> None:
    for line in proc.stdout:                     # type: ignore[union-attr]
        sink.write("[%s] %s" % (tag, line))
        sink.flush()


def launch(script: str, script_args: Sequence[str], num_workers: int, num_ps: int = 1,
           protocol: str = "grpc", strategy: Optional[str] = None,
           timeout_s: Optional[float] = None, sink=None) -> int:
    sink = sink or sys.stdout
    cluster, tasks = plan_tasks(num_workers, num_ps, protocol)
    procs: List[Tuple[str, subprocess.Popen]] = []
    pumps: List[threading.Thread] = []
    for job, index in tasks:
        proc = subprocess.Popen(
            child_command(script, script_args, cluster, job, index, protocol, strategy),
            env=child_env(cluster, job, index, protocol), stdout=subprocess.PIPE,
            stderr=subprocess.STDOUT, text=True, bufsize=1)
        tag = "%s:%d" % (job, index)
        procs.append((tag, proc))
        t = threading.Thread(target=_pump, args=(proc, tag, sink), daemon=True)
        t.start()
        pumps.append(t)
    deadline = None if timeout_s is None else time.monotonic() + timeout_s
    status = 0
    try:
        pending = dict(procs)
        while pending:
            for tag, proc in list(pending.items()):
                rc = proc.poll()
                if rc is None:
                    continue
                del pending[tag]
                if rc != 0 and status == 0:
                    status = rc
            if status != 0:
                break
            # only ps tasks left and nobody will tell them to stop -> we are done
            if pending and all(tag.startswith("ps:") for tag in pending) and protocol != "grpc":
                break
            if deadline is not None and time.monotonic() > deadline:
                status = 124
                break
            time.sleep(0.05)
    finally:
        for tag, proc in procs:
            if proc.poll() is None:
                proc.send_signal(signal.SIGTERM)
        for tag, proc in procs:
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                proc.kill()
                proc.wait()
        for t in pumps:
            t.join(timeout=2)
    return status


def count_gpus() -> int:
    """GPUs this job may use, found WITHOUT a HIP call: the launcher is the parent of one process
    per GPU and must not hold a GPU context (or a forked runtime) of its own. The kernel driver
    lists every compute node under ``/sys/class/kfd/kfd/topology/nodes``; GPUs are the nodes with
    SIMDs (CPUs have ``simd_count 0``). ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` /
    ``CUDA_VISIBLE_DEVICES`` narrow the count the way they narrow the runtime's view."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    found = 0
    try:
        for name in os.listdir(nodes):
            try:
                with open(os.path.join(nodes, name, "properties")) as fh:
                    props = dict(line.split(None, 1) for line in fh if " " in line)
            except OSError:
                continue
            if int(props.get("simd_count", "0").strip() or 0) > 0:
                found += 1
    except OSError:
        found = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        value = os.environ.get(var)
        if value is not None:
            listed = [v for v in value.split(",") if v.strip() != ""]
            if listed and listed[0].strip() == "-1":
                listed = []
            found = min(found, len(listed)) if found else 0
    return found


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="distributed_tensorflow_amd.launch", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--num_workers", type=int, default=0,
                    help="worker processes (default: one per visible GPU, at least 1)")
    ap.add_argument("--num_ps", type=int, default=1, help="parameter-server tasks (grpc only)")
    ap.add_argument("--protocol", choices=("grpc", "rccl"), default="grpc")
    ap.add_argument("--strategy", default=None, help="rccl: allreduce_sync | reduce_scatter_sync")
    ap.add_argument("--timeout", type=float, default=None, help="kill the job after this many seconds")
    ap.add_argument("script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)
    workers = args.num_workers
    if workers <= 0:
        workers = max(1, count_gpus())
    extra = [a for a in args.script_args if a != "--"]
    return launch(args.script, extra, workers, args.num_ps, args.protocol, args.strategy,
                  args.timeout)


if __name__ == "__main__":
    sys.exit(main())
