"""Communicators for the one exchange step of the path: the all-reduce of the packed
expected sufficient statistics ``[A_raw | xbar | neff | S | lb]`` before the global
natural-gradient step (the ``A_inter += A_i ; emit_inter[k] += e_i[k]`` accumulation
of reference hmmsgd_metaobs.py:430-436 extended across GPUs).

* ``RcclComm``  -- production: ncclAllReduce(sum, fp64) on the device buffer over
  RCCL/xGMI through the C ABI (``svihmm_allreduce_packed``); one process per GPU.
* ``TorchDistComm`` -- host all-reduce through an initialised ``torch.distributed``
  process group (gloo); used by the CPU multi-process tests of the sharding logic.
Both give every rank bit-identical statistics, so the deterministic host global step
keeps the replicas in lock-step without a parameter broadcast.
"""
import numpy as np

from .engine import PackedStats


class RcclComm(object):
    def __init__(self, engine, rank, size, exchange_uid):
        """``exchange_uid(uid_or_None) -> uid``: host-side broadcast of the 128-byte
        ncclUniqueId from rank 0 (e.g. via torch.distributed gloo or a TCP store)."""
        self.rank, self.size = int(rank), int(size)
        uid = exchange_uid(engine.comm_unique_id() if self.rank == 0 else None)
        engine.comm_init(uid, self.rank, self.size)

    def allreduce_stats(self, engine, K, D):
        engine.allreduce_packed()
        return engine.read_packed()

    def barrier(self, engine):
        engine.allreduce_host(np.zeros(1))


class TorchDistComm(object):
    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def allreduce_inplace(self, buf):
        import torch
        t = torch.from_numpy(buf)
        self._dist.all_reduce(t, group=self.group)
        return buf

    def allreduce_stats(self, engine, K, D):
        st = engine.read_packed()
        buf = np.ascontiguousarray(st.buf)
        self.allreduce_inplace(buf)
        return type(st)(buf, K, getattr(st, "V", D))

    def barrier(self, engine=None):
        self._dist.barrier(group=self.group)


def file_uid_exchange(rank, tag=None, timeout=300.0, directory=None):
    """Single-node rendezvous without torch: rank 0 publishes the 128-byte ncclUniqueId
    in a file named after the launcher's pid (all workers of one ``torch.distributed.run``
    share the parent pid) and MASTER_PORT; the other ranks poll for it.  Returns a
    function suitable as ``exchange_uid`` of ``RcclComm``."""
    import os
    import tempfile
    import time
    if tag is None:
        tag = "%s_%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid(),
                            os.environ.get("MASTER_PORT", "0"))
    directory = directory or tempfile.gettempdir()
    path = os.path.join(directory, "svihmm_uid_%s.bin" % tag)

    def exchange(uid):
        if rank == 0:
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)          # atomic publish
            return uid
        t0 = time.time()
        while True:
            try:
                # a file left behind by a crashed earlier job with the same tag is older than
                # this job's workers (which all start within seconds of each other)
                if os.path.getmtime(path) >= t0 - 120.0:
                    with open(path, "rb") as f:
                        data = f.read()
                    if len(data) == 128:
                        return data
            except OSError:
                pass
            if time.time() - t0 > timeout:
                raise RuntimeError("rendezvous timeout waiting for %s" % path)
            time.sleep(0.01)

    exchange.path = path
    return exchange


def torch_uid_exchange(uid):
    """broadcast the ncclUniqueId over an initialised torch.distributed group."""
    import torch.distributed as dist
    box = [uid]
    dist.broadcast_object_list(box, src=0)
    return box[0]
