"""pysvihmm_amd -- MI355X-native SVI-HMM E-step engine behind the class surface of
dillonalaird/pysvihmm (``hmmbatchcd.VBHMM``, ``hmmbatchsgd.VBHMM``,
``hmmsgd_metaobs.VBHMM``, ``VariationalHMMBase``, pybasicbayes-style emission plugins).

The hot path (emission expected log-likelihood, forward/backward, posteriors,
expected sufficient statistics) runs as hand-written HIP kernels for gfx950 behind
the C ABI of ``include/svihmm.h`` (``libsvihmm_hip.so``, loaded with ctypes).
"""
__version__ = "0.1.0"
