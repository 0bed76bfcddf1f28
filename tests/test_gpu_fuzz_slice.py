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
    for i in range(8):
        fuzz_gpu.run_case(eng, L, ref_c, fuzz_gpu.draw_case(rng), 3100000 + i)


def test_api_and_class_cases(eng):
    from tests import fuzz_gpu
    from pysvihmm_amd import _lib as L
    eng.set_precision("f64")
    for i in range(4):
        fuzz_gpu.run_api_case(eng, L, 3170000 + i)
    for i in range(4):
        fuzz_gpu.run_class_case(eng, 3190000 + i)


# ---- the campaign itself, time-boxed: every leg of tests/fuzz_gpu.py through its
# own driver (main: case drawing, failure accounting, engine re-creation after a hard error), fixed
# seed, 8 s per leg = 40 s in all.  A failure prints the leg's seed for `fuzz_gpu.py --replay`.
LEGS = {"statistics": ["--cases", "100000", "--chains", "0"],
        "chains": ["--cases", "0", "--chains", "1000"],
        "api": ["--cases", "0", "--chains", "0", "--api", "100000"],
        "sequences": ["--cases", "0", "--chains", "0", "--sequences", "100000"],
        "classes": ["--cases", "0", "--chains", "0", "--classes", "100000"]}


@pytest.mark.parametrize("leg", sorted(LEGS))
def test_campaign_leg_time_boxed(leg, capsys):
    from tests import fuzz_gpu
    rc = fuzz_gpu.main(["--seed", "404", "--seconds", "8"] + LEGS[leg])
    out = capsys.readouterr().out
    assert rc == 0, out[-3000:]
    import re
    m = re.search(r"fuzz: (\d+) cases, (\d+) failures", out)
    assert m and int(m.group(1)) >= 3 and int(m.group(2)) == 0, out[-1000:]
