import numpy as np, sys
sys.path.insert(0, '.')
from krylov_amd import capi, problems as P
for bc in [(0,0,1,1,0,1), (1,1,1,1,1,1), (0,1,0,1,1,0)]:
    for L in (1, 2):
        p = P.smooth_var(24, bc=bc) if 'bc' in P.smooth_var.__code__.co_varnames else None
        if p is None:
            p = P.smooth_var(24); p.bc = bc
        a = capi.solve_problem(p, L=L, eps=1e-8, max_outer=30)
        b = capi.solve_problem(p, L=L, eps=1e-8, max_outer=30, flags=1)
        ha, hb = np.array(a['history']), np.array(b['history'])
        n = min(len(ha), len(hb), 6)
        print(bc, L, 'fused', a['fused'], b['fused'], 'iters', len(ha), len(hb),
              'hist rel', np.max(np.abs(ha[:n]-hb[:n])/np.abs(hb[:n])),
              'phi diff', np.max(np.abs(a['phi']-b['phi'])), 'status', a['status'], b['status'])
