"""The C-ABI library loads and exports every symbol include/svihmm.h declares, and the
ctypes prototypes cover exactly that set (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from pysvihmm_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _functions_of(header):
    src = open(os.path.join(REPO, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svihmm_[a-z0-9_]+)\s*\(", src)))


def declared_functions():
    """The drop-in boundary (svihmm.h) + the measurement / test hooks (svihmm_debug.h)."""
    return sorted(set(_functions_of("svihmm.h")) | set(_functions_of("svihmm_debug.h")))


def test_measurement_hooks_are_not_in_the_product_header():
    product = set(_functions_of("svihmm.h"))
    hooks = set(_functions_of("svihmm_debug.h"))
    assert hooks == {"svihmm_profile_enable", "svihmm_profile_reset", "svihmm_profile_read",
                     "svihmm_kernel_name", "svihmm_last_kernel_name", "svihmm_set_variant",
                     "svihmm_svi_recoveries", "svihmm_selftest_mfma"}
    assert not (product & hooks)


def test_header_symbols_exported():
    names = declared_functions()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "missing export: " + n


def test_ctypes_prototypes_match_header():
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_load_and_version_without_gpu():
    lib = _lib.load()
    assert lib.svihmm_abi_version() == _lib.ABI_VERSION == 3
    assert lib.svihmm_packed_size(64, 32) == 64 * 64 + 64 * 32 + 64 + 64 * 32 * 32 + 1
    assert lib.svihmm_kernel_name(0) == b"emission"
