"""Test-only communicator: host all-reduce through an initialised ``torch.distributed``
process group (gloo), same protocol as ``pysvihmm_amd.comm.RcclComm``; used by the CPU
multi-process tests of the sharding logic (world size 2 and 3)."""
import numpy as np


class TorchDistComm(object):
    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def bind_engine(self, engine):
        """SVI loop inside the (oracle) engine: it all-reduces its packed statistics through us."""
        engine._comm = self

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


def torch_uid_exchange(uid):
    """broadcast the ncclUniqueId over an initialised torch.distributed group."""
    import torch.distributed as dist
    box = [uid]
    dist.broadcast_object_list(box, src=0)
    return box[0]
