"""Communicators for the one exchange step of the path: the all-reduce of the packed
expected sufficient statistics ``[A_raw | xbar | neff | S | lb]`` before the global
natural-gradient step (the ``A_inter += A_i ; emit_inter[k] += e_i[k]`` accumulation
of reference hmmsgd_metaobs.py:430-436 extended across GPUs).

``RcclComm``: ncclAllReduce(sum, fp64) on the device buffer over RCCL/xGMI through the C ABI
(``svihmm_allreduce_packed``); one process per GPU, no PyTorch in the process.  Every rank
ends up with bit-identical statistics, so the deterministic host global step keeps the
replicas in lock-step without a parameter broadcast.  The ncclUniqueId travels through a
file (``file_uid_exchange``) or any callable the caller supplies.  (The CPU multi-process
tests of the sharding logic use a gloo communicator with the same protocol; it lives with
the tests, ``tests/dist_helpers.py``.)
"""
import numpy as np


class RcclComm(object):
    def __init__(self, engine, rank, size, exchange_uid):
        """``exchange_uid(uid_or_None) -> uid``: host-side broadcast of the 128-byte
        ncclUniqueId from rank 0 (``file_uid_exchange``, a TCP store, ...)."""
        self.rank, self.size = int(rank), int(size)
        uid = exchange_uid(engine.comm_unique_id() if self.rank == 0 else None)
        engine.comm_init(uid, self.rank, self.size)      # collective: returns once every rank has joined
        if self.rank == 0 and getattr(exchange_uid, "path", None):
            # consumed: a later job with the same tag (one long-lived parent launching several jobs on
            # one MASTER_PORT) must never find this job's id
            import os
            try:
                os.remove(exchange_uid.path)
            except OSError:
                pass

        self._engine = engine

    def bind_engine(self, engine):
        """The device-resident SVI loop all-reduces inside the engine (on the handle's stream):
        the communicator must be the one this engine's handle was initialised with."""
        if engine is not self._engine:
            raise RuntimeError("RcclComm is bound to another engine handle")

    def allreduce_stats(self, engine, K, D):
        engine.allreduce_packed()
        return engine.read_packed()

    def barrier(self, engine):
        engine.allreduce_host(np.zeros(1))


def file_uid_exchange(rank, tag=None, timeout=None, directory=None):
    """Single-node rendezvous without torch: rank 0 publishes the 128-byte ncclUniqueId
    in a file named after the launcher's pid (all workers of one ``torch.distributed.run``
    share the parent pid) and MASTER_PORT; the other ranks poll for it.  Returns a
    function suitable as ``exchange_uid`` of ``RcclComm``."""
    import os
    import tempfile
    import time
    if timeout is None:     # (bounded by default: SVIHMM_RENDEZVOUS_TIMEOUT seconds, 120 when unset)
        timeout = float(os.environ.get("SVIHMM_RENDEZVOUS_TIMEOUT", "120"))
    if tag is None:
        tag = "%s_%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid(),
                            os.environ.get("MASTER_PORT", "0"))
    directory = directory or tempfile.gettempdir()
    path = os.path.join(directory, "svihmm_uid_%s.bin" % tag)

    def launcher_start():
        """Wall-clock start of the parent (launcher) process; None if /proc does not say."""
        try:
            with open("/proc/%d/stat" % os.getppid()) as f:
                ticks = float(f.read().rsplit(")", 1)[1].split()[19])      # field 22: starttime
            with open("/proc/stat") as f:
                btime = [float(l.split()[1]) for l in f if l.startswith("btime")][0]
            return btime + ticks / os.sysconf("SC_CLK_TCK")
        except (OSError, ValueError, IndexError):
            return None

    def exchange(uid):
        if rank == 0:
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)          # atomic publish
            return uid
        t0 = time.time()
        # a file left behind by a crashed earlier job with the same tag is older than this
        # job's launcher (fallback: older than two minutes before this worker got here)
        # (no bound relative to THIS worker's arrival: workers may start minutes apart.  A job that
        #  completes its rendezvous removes the file -- RcclComm -- so what a long-lived launcher's
        #  earlier jobs can leave behind is limited to jobs that crashed inside the rendezvous)
        born = launcher_start()
        fresh = (born - 1.0) if born is not None else (t0 - 120.0)
        while True:
            try:
                if os.path.getmtime(path) >= fresh:
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
