"""Device-side synthetic sequences (reference gen_synthetic.py:27-44): exact
agreement with the NumPy restatement on the same counter-based random stream, the statistics
of the process, and use as the resident observation copy of an E-step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(K, D, seed):
    rng = np.random.default_rng(seed)
    tran = rng.dirichlet(np.ones(K) * 0.5, size=K) * 0.3 + 0.7 * np.eye(K)
    means = rng.normal(0, 5, size=(K, D))
    a = rng.normal(size=(K, D, D + 2))
    cov = np.einsum('kij,klj->kil', a, a) / D + 0.2 * np.eye(D)
    return tran, means, np.linalg.cholesky(cov), cov


@pytest.mark.parametrize("K,D,T", [(3, 2, 700), (16, 8, 5000), (64, 5, 70001), (40, 33, 3000)])
def test_generate_equals_counter_based_oracle(K, D, T):
    from pysvihmm_amd.engine import HipEngine
    from oracle import ref_numpy as R
    tran, means, chols, _ = _model(K, D, K + D)
    e = HipEngine(0)
    e.generate(tran, means, chols, T, seed=0x1234567890ABCDEF + K)
    obs, sts = e.read_generated()
    ro, rz = R.generate_counter_based(tran, means, chols, T, 0x1234567890ABCDEF + K)
    assert sts[0] == 0 and np.array_equal(sts, rz)
    np.testing.assert_allclose(obs, ro, rtol=1e-12, atol=1e-12)
    # reproducible, and a different seed gives a different sequence
    e.generate(tran, means, chols, T, seed=0x1234567890ABCDEF + K)
    obs2, sts2 = e.read_generated()
    assert np.array_equal(sts, sts2) and np.array_equal(obs, obs2)
    e.generate(tran, means, chols, T, seed=7)
    assert not np.array_equal(e.read_generated(want_obs=False)[1], sts)
    e.close()


def test_generated_process_statistics_and_estep():
    """T = 2e6: empirical transition frequencies and per-state moments match the model; the
    sequence is the engine's resident observation copy (E-step runs on it directly)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import gen_synthetic
    from scipy.special import digamma
    K, D, T = 8, 4, 2000000
    tran, means, chols, cov = _model(K, D, 3)
    obs, sts, e = gen_synthetic.generate_data_device(tran, means, chols, T, seed=11, want_obs=True)
    cnt = np.zeros((K, K)); np.add.at(cnt, (sts[:-1], sts[1:]), 1)
    emp = cnt / cnt.sum(1, keepdims=True)
    assert np.abs(emp - tran).max() < 5 * np.sqrt(0.25 / cnt.sum(1).min())
    for k in range(K):
        x = obs[sts == k]
        se = np.sqrt(np.diag(cov[k]) / len(x))
        assert np.all(np.abs(x.mean(0) - means[k]) < 6 * se)
        np.testing.assert_allclose(np.cov(x.T), cov[k], rtol=0.05, atol=0.05)
    # E-step on the resident sequence == E-step after uploading the same observations
    vt = 1.0 + cnt[:K, :K] / 1000.0
    ltran = digamma(vt) - digamma(vt.sum(1))[:, None]
    mod_init = np.log(np.full(K, 1.0 / K))
    e.set_globals(mod_init, ltran)
    e.set_emission_niw(means, cov * 3.0, np.full(K, 1.0), np.full(K, D + 3.0))
    starts = np.arange(0, T - 300, 9973)
    a = e.estep(starts, 257).buf.copy()
    e.set_obs(obs, None)
    b = e.estep(starts, 257).buf
    # (not bit-equal: the handle keeps either copy centred on a point of its own choosing --
    #  average state mean for the generator, sample mean for an upload)
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-8)
    e.close()
