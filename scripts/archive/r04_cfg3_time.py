"""cfg3 (262 144 tips, JTT k = 20, joint sweep) timed as bench.py's secondary does (graph replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastml_amd import hip, synthetic
from pastml_amd.models.JTTModel import JTT_FREQUENCIES, JTT_RATE_MATRIX
from pastml_amd.models.generator import get_diagonalisation
flat = synthetic.balanced_forest(18)
d, A, Ainv = get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
spec = dict(kind=2, pi=JTT_FREQUENCIES, d=d, A=A, Ainv=Ainv)
with hip.Engine(flat, 1, 20) as eng:
    eng.set_models([(spec, (1.0, 0.0, 1.0))])
    eng.set_tip_states(synthetic.tip_states(flat.n_tips, 20, 0))
    def timed(fn, reps=50):
        fn(); eng.sync(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        eng.sync(); return (time.perf_counter() - t0) / reps * 1e3
    res = []
    for _ in range(3):
        res.append((timed(lambda: eng.bottom_up(False)), timed(lambda: eng.joint_pass(copy_out=False)), timed(lambda: eng.marginal_pass(posterior=False, lh=False))))
    lnl = eng.bottom_up(False)[0]
print('joint sweep %.4f ms, joint pass %.4f ms, marginal pass %.4f ms  (best of 3)  lnL %.6f' % (min(r[0] for r in res), min(r[1] for r in res), min(r[2] for r in res), lnl))
