"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo job running
the sharded SVI minibatch E-step (host logic + all-reduce; oracle engine on CPU)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out_dir, fixture = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    dist.init_process_group("gloo")
    from oracle.engine import OracleEngine
    from pysvihmm_amd import hmmsgd_metaobs
    from tests.dist_helpers import TorchDistComm
    from tests.test_host_logic import emit_from_fixture
    g = np.load(fixture)
    K = int(g["K"])
    comm = TorchDistComm()
    hmm = hmmsgd_metaobs.VBHMM(
        g["obs"].copy(), np.ones(K), g["prior_tran"], emit_from_fixture(g, K),
        tau=float(g["tau"]), kappa=float(g["kappa"]), metaobs_half=int(g["L"]),
        mb_sz=int(g["S"]), mask=g["mask"], init_tran=g["init_tran"], maxit=int(g["maxit"]),
        seed=int(g["seed"]), engine=OracleEngine(), comm=comm)
    hmm.infer()
    np.savez(os.path.join(out_dir, "rank%d.npz" % comm.rank), var_tran=hmm.var_tran,
             elbo=hmm.elbo_vec, mu=np.array([e.mu_mf for e in hmm.var_emit]),
             sigma=np.array([e.sigma_mf for e in hmm.var_emit]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
