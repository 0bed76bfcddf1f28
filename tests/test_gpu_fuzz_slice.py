"""A fixed-seed slice of the randomised campaign (tests/fuzz_gpu.py) inside the collected suite:
the call-sequence leg -- random API calls on ONE handle, now including a class borrowing the
handle, explicit shifts of the centre, re-uploads under live parameters, data far from the origin
and the diagonal family -- plus a few cases of the statistics / API / class legs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pysvihmm_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


@pytest.mark.parametrize("seed", [300001, 300002, 300003, 300004, 300005, 300006])
def test_call_sequences_on_one_handle(eng, seed):
    from tests import fuzz_gpu
    from pysvihmm_amd import _lib as L
    eng.set_precision("f64")
    hist = fuzz_gpu.run_sequence(eng, L, seed, nops=30)
    assert len(hist) == 30


def test_statistics_cases(eng):
    from tests import fuzz_gpu
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    eng.set_precision("f64")
    rng = np.random.default_rng(31)
    for i in range(12):
        fuzz_gpu.run_case(eng, L, ref_c, fuzz_gpu.draw_case(rng), 3100000 + i)


def test_api_and_class_cases(eng):
    from tests import fuzz_gpu
    from pysvihmm_amd import _lib as L
    eng.set_precision("f64")
    for i in range(4):
        fuzz_gpu.run_api_case(eng, L, 3170000 + i)
    for i in range(4):
        fuzz_gpu.run_class_case(eng, 3190000 + i)
