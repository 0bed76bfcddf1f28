import sys; sys.path.insert(0,'.')
from pysvihmm_amd.engine import HipEngine
e=HipEngine(0)
for w,n in ((400,'window-major (16 x 512 B per step)'),(401,'step-major (8 KB per step)')):
    print('%-40s %.2f TB/s (1 GB read + 1 GB write)'%(n,e.peak_fp64(w)))
