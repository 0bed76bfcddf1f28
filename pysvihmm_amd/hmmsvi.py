"""``hmmsvi`` -- name compatibility only.

The reference's ``hmmsvi.SVIHMM`` (hmmsvi.py:21) is a stale skeleton that cannot be
constructed or run (wrong ``super().__init__`` argument order hmmsvi.py:62-63,
``np.squash`` hmmsvi.py:139,148, undefined ``self.N`` hmmsvi.py:193; SURVEY.md
quirk Q12).  BASELINE.json's ``HMMSVI`` / the reference's ``SVIHMM`` therefore
resolve to the working SVI implementation, ``hmmsgd_metaobs.VBHMM``.
"""
from .hmmsgd_metaobs import VBHMM, MetaObs  # noqa: F401

SVIHMM = VBHMM
HMMSVI = VBHMM
