"""N>1 path on CPU: world_size-2 (and 3) gloo jobs.  Windows of each minibatch are
sharded round-robin over the ranks, the packed statistics are all-reduced, every rank
applies the identical host global step.  The result must equal the single-process
reference trace and be identical on every rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_minibatch_allreduce(tmp_path, world):
    fixture = os.path.join(GOLDEN, "metaobs_K4_D2_L10_mask.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()),
           os.path.join(REPO, "tests", "_dist_worker.py"), str(tmp_path), fixture]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = np.load(fixture)
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % i)) for i in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["var_tran"], g["it_var_tran_new"][-1], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(o["mu"], g["it_new_mu"][-1], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(o["sigma"], g["it_new_sigma"][-1], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(o["elbo"], g["elbo_vec"], rtol=1e-9)
    for o in outs[1:]:                      # replicas stay in lock-step bit for bit
        np.testing.assert_array_equal(o["var_tran"], outs[0]["var_tran"])
        np.testing.assert_array_equal(o["sigma"], outs[0]["sigma"])


def _uid_worker(rank, tag, directory, q):
    from pysvihmm_amd.comm import file_uid_exchange
    ex = file_uid_exchange(rank, tag=tag, directory=directory, timeout=30)
    uid = bytes(range(128)) if rank == 0 else None
    if rank == 0:
        import time
        time.sleep(0.3)                      # late publisher: the others must poll
    q.put((rank, ex(uid)))


def test_file_rendezvous_of_unique_id(tmp_path):
    """bench.py's torch-free rendezvous: rank 0 publishes the 128-byte ncclUniqueId
    atomically, the other ranks poll for it."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_uid_worker, args=(r, "t%d" % os.getpid(), str(tmp_path), q))
          for r in range(4)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
    assert all(got[r] == bytes(range(128)) for r in range(4))
