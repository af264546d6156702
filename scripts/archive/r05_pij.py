"""P(t) batch (pml_pij_batch, eigen models): kernel time by HIP events for a few k on the 262 144-tip balanced tree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastml_amd import hip, synthetic
from pastml_amd.models.generator import get_diagonalisation
from pastml_amd.models.JTTModel import JTT_FREQUENCIES, JTT_RATE_MATRIX

ks_list = [int(a) for a in sys.argv[1:]] or [20, 32, 17, 24]
flat = synthetic.balanced_forest(18)
rng = np.random.default_rng(11)
for k in ks_list:
    if k == 20:
        pi, (d, A, Ainv) = JTT_FREQUENCIES, get_diagonalisation(JTT_FREQUENCIES, JTT_RATE_MATRIX)
    else:
        pi = rng.dirichlet(np.ones(k) * 4)
        r = np.triu(rng.uniform(0.1, 2.0, size=(k, k)), 1)
        d, A, Ainv = get_diagonalisation(pi, r + r.T)
    spec = dict(kind=2, pi=pi, d=d, A=A, Ainv=Ainv)
    with hip.Engine(flat, 1, k) as eng:
        eng.set_models([(spec, (1.0, 0.0, 1.0))])
        eng.pij_batch()
        eng.sync()
        eng.profile_enable(True)
        eng.profile_read(2, reset=True)
        for _ in range(20):
            eng.set_models([(spec, (1.0, 0.0, 1.0))])
            eng.pij_batch()
        eng.sync()
        ms, n = eng.profile_read(2, reset=True)
        ms /= n
        kp = 4 * ((k + 3) // 4)
        gb = flat.n_nodes * 8.0 * k * kp / 1e9
        print('k=%d stage_rows=%s: %.4f ms per batch, %.2f GB out, %.0f GB/s = %.3f of the HBM peak; %.1f TFLOP/s (2 k^3 per branch)'
              % (k, os.environ.get('PASTML_HIP_PIJ_STAGE_ROWS', 'default'), ms, gb, gb / ms * 1e3, gb / ms * 1e3 / 8000,
                 2.0 * k ** 3 * flat.n_nodes / ms / 1e9))
