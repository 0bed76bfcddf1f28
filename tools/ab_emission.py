import sys,time; sys.path.insert(0,'.')
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
from oracle import ref_c
K,D,Lm,T=64,32,257,1000000
pb=make_problem(K,D,T,seed=1,sep=5.0)
e=HipEngine(0); e.set_obs(pb['obs'],None); e.set_globals(pb['mod_init'],pb['ltran']); e.set_emission_niw(pb['mu'],pb['sigma'],pb['kappa'],pb['nu'])
B=3891; starts=np.arange(B,dtype=np.int64)*Lm
ref=ref_c.lliks_niw(pb['obs'][:Lm],pb['mu'],pb['sigma'],pb['kappa'],pb['nu'])
for rnd in range(3):
  for mt in (2,4):
    e.set_variant('emission_mt',mt)
    e.estep(starts,Lm,read=False); e.sync()
    e.profile(True); e.profile_reset()
    for _ in range(5): e.estep(starts,Lm,read=False)
    p=e.profile_read(); e.profile(False)
    ll=e.read_rows('lliks',0,Lm)
    print(rnd,'MT',mt,'emission %.3f ms'%(p['emission'][0]/p['emission'][1]),'err',np.abs(ll-ref).max())
