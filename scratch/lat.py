import sys, time; sys.path.insert(0,'.')
import numpy as np
from pastml_amd import hip
from pastml_amd.tree import FlatForest
def timed(fn, reps=300):
    fn(); t0=time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter()-t0)/reps*1e6
rng=np.random.default_rng(0)
for tips,k in ((154,5),(3619,12),(3619,2)):
    flat=FlatForest.random(tips, seed=1, max_arity=2)
    spec=dict(kind=0, pi=rng.dirichlet(np.ones(k)))
    eng=hip.Engine(flat,1,k); eng.set_tip_states(rng.integers(0,k,size=flat.n_tips)); eng.set_models([(spec,(1.0,0.0,1.0))])
    print(tips,k,'levels',flat.n_bu_levels,'set_models us %.1f'%timed(lambda: eng.set_models([(spec,(1.0,0.0,1.0))])), 'bottom_up(no model change) us %.1f'%timed(lambda: eng.bottom_up(True)), 'both us %.1f'%timed(lambda: (eng.set_models([(spec,(1.0,0.0,1.0))]), eng.bottom_up(True))), 'td us %.1f'%timed(lambda: eng.top_down_marginals(posterior=False, lh=False)), 'sync us %.1f'%timed(lambda: eng.sync()))
    eng.close()
