"""round 6: wall-clock stamps of one fused sweep + statistics launch (SVIHMM_PIPE_DBG): when the sweep workgroups run,
when each statistics stage's gate opens and when its k-steps end."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _workload import bench_problem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = HipEngine(0)
pb = bench_problem(e)
LM, T = bench.LM, bench.T
st = (np.arange(B, dtype=np.int64) * (T // B)) % (T - LM)
if len(sys.argv) > 2:
    e.set_variant("pipeline", int(sys.argv[2]))
e.set_globals(pb["mod_init"], pb["ltran"]); e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
for _ in range(5):
    e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
e.sync()
path = "/tmp/pipe_dbg.txt"
os.environ["SVIHMM_PIPE_DBG"] = path
e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
e.sync()
del os.environ["SVIHMM_PIPE_DBG"]
lines = open(path).read().splitlines()
print(lines[0])
hdr = lines[0].split()
nsw, NS = int(hdr[2]), int(hdr[8])
raw = [[float(x) for x in l.split()[1:]] for l in lines[1:]]
W = max(len(r) for r in raw)
rows = np.array([r + [-1.0] * (W - len(r)) for r in raw])
sw = rows[:nsw]
if sw.shape[1] >= 13:
    print("sweep wave 0 of each workgroup, mean stamps (us): step 63 / 127 / 191 / 255, chain done, function end")
    for name, sel in (("fwd", sw[0::2]), ("bwd", sw[1::2])):
        print("  %s: %s | %.1f | %.1f" % (name, " ".join("%.1f" % sel[:, 2 + 2 * q].mean() for q in range(4)), sel[:, 12].mean(), sel[:, 1].mean()))
print("sweep workgroups: begin %.1f .. %.1f us, end %.1f .. %.1f us (fwd even / bwd odd: fwd end mean %.1f, bwd end mean %.1f)" % (
    sw[:, 0].min(), sw[:, 0].max(), sw[:, 1].min(), sw[:, 1].max(), sw[0::2, 1].mean(), sw[1::2, 1].mean()))
stt = rows[nsw:]
if stt.shape[1] >= 32 and stt[:, 29].max() > 0:
    v = lambda c: stt[:, c][stt[:, c] > 0]
    if stt[:, 23].max() > 0:
        print("second tile of a workgroup, mean stamps: rows + previous stores retired %.2f | arrival of the previous tile sent %.2f | k-steps done %.2f | stores issued %.2f" % (
            v(23).mean(), v(30).mean(), v(26).mean(), v(12).mean()))
    print("emission role of the statistics workgroups: begin %.1f .. %.1f, first tile done %.1f .. %.1f (mean %.1f), second %.1f .. %.1f (mean %.1f), all rounds done %.1f .. %.1f (mean %.1f)" % (
        v(28).min(), v(28).max(), v(30).min(), v(30).max(), v(30).mean(), v(31).min(), v(31).max(), v(31).mean(), v(29).min(), v(29).max(), v(29).mean()))
print("statistics workgroups: begin %.1f .. %.1f" % (stt[:, 0].min(), stt[:, 0].max()))
print("  band 0 open (first rows requested): %.1f .. %.1f (mean %.1f)" % (stt[:, 1].min(), stt[:, 1].max(), stt[:, 1].mean()))
for s in range(NS):
    d = stt[:, 2 + 2 * s]
    line = "  stage %d: k-steps end %.1f .. %.1f (mean %.1f)" % (s, d.min(), d.max(), d.mean())
    if s + 1 < NS and stt.shape[1] > 15 + 2 * s:
        line += "; rows of stage %d requested at %.1f (mean), committed at %.1f" % (s + 1, stt[:, 14 + 2 * s].mean(), stt[:, 15 + 2 * s].mean())
    print(line)
e.close()
