"""Worker for tests/test_gpu_rccl.py: one rank (= one GPU) of a world_size-N RCCL job running the
sharded SVI minibatch E-step on the HIP engine; no torch in the process (file rendezvous of the
ncclUniqueId, RANK / LOCAL_RANK / WORLD_SIZE from the environment)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out_dir, fixture = sys.argv[1], sys.argv[2]
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    from pysvihmm_amd import hmmsgd_metaobs
    from pysvihmm_amd.comm import RcclComm, file_uid_exchange
    from pysvihmm_amd.engine import HipEngine
    from tests.test_host_logic import emit_from_fixture
    g = np.load(fixture)
    K = int(g["K"])
    eng = HipEngine(int(os.environ.get("LOCAL_RANK", rank)))
    comm = RcclComm(eng, rank, world, file_uid_exchange(rank, tag=os.environ["SVIHMM_TEST_TAG"], directory=out_dir))
    assert eng.comm_count() == world
    hmm = hmmsgd_metaobs.VBHMM(
        g["obs"].copy(), np.ones(K), g["prior_tran"], emit_from_fixture(g, K),
        tau=float(g["tau"]), kappa=float(g["kappa"]), metaobs_half=int(g["L"]),
        mb_sz=int(g["S"]), mask=g["mask"], init_tran=g["init_tran"], maxit=int(g["maxit"]),
        seed=int(g["seed"]), engine=eng, comm=comm)
    hmm.infer()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), var_tran=hmm.var_tran, elbo=hmm.elbo_vec,
             mu=np.array([e.mu_mf for e in hmm.var_emit]), sigma=np.array([e.sigma_mf for e in hmm.var_emit]),
             nranks=eng.comm_count())
    comm.barrier(eng)
    eng.close()


if __name__ == "__main__":
    main()
