"""fp64 MFMA / FMA throughput probes (roofline calibration); tools-only library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _probe import probe_fp64 as peak
for w, name in ((0, 'mfma_f64 16x16x4, 8 blocks/CU, 4 chains'), (3, 'mfma 2 blocks/CU'), (2, 'mfma 1 block/CU'), (1, 'v_fma_f64')):
    print('%-42s %.1f TFLOP/s' % (name, peak(w)))
for n in (1, 2, 4, 8, 12, 16):
    ns = peak(100 + n)
    print('1 wave/SIMD, %2d independent accumulators: %.1f ns per MFMA per SIMD -> %.1f TFLOP/s chip' % (n, ns, 2048.0 / ns * 1024 / 1e3))
print('MFMA / fp64-VALU overlap probe (2 waves/SIMD; ns per iteration of 8 MFMA [+ 8*NF v_fma_f64]):')
for code, name in ((201, '8 MFMA only'), (280, '64 FMA only'), (281, '8 MFMA + 64 FMA'), (360, '128 FMA only'), (361, '8 MFMA + 128 FMA')):
    print('  %-20s %.1f ns' % (name, peak(code)))
