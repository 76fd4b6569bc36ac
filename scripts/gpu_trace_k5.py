import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S, workloads as W, optimizer as HO, _lib
from hortimapping_amd.decoder import DecoderWeights
L, B = 256, 8
params = S.make_synthetic_decoder(L, seed=2, r0=0.04, aniso=(1.0, 0.75, 1.3))
dec = DecoderWeights.from_params(params); dec.set_precision('f16x3')
dicts = W.make_c2_instances(params, dec, list(range(B)), kind="joint")
insts = [W.to_instance(d) for d in dicts]
tr = torch.zeros(32, dtype=torch.int64, device='cuda')
lib = _lib.lib(); lib.hm_debug_set_k5_trace.argtypes = [ctypes.c_void_p]
lib.hm_debug_set_k5_trace(tr.data_ptr())
res = HO.optimize_batch(dec, W.c2_opt_cfg(max_iter=3), insts)
torch.cuda.synchronize()
t = tr.cpu().numpy()
import os
if t[7] == 1:
    print(f"assemble {t[1]-t[0]:8d} ticks\nPCG solve {t[2]-t[1]:8d} ticks (fast path taken)\ntotal to end of solve {t[6]-t[0]:8d} ticks")
    print(f"  A -> LDS {t[18]-t[17]:8d}; iterations pass0 {t[25]} pass1 {t[26]}; fp64 true residual {t[24]-t[23]:8d}")
    print(f"  iteration 1 of pass 0: row loop {t[20]-t[19]:6d} | shuffles+part+barrier {t[21]-t[20]:6d} | combine+sum(pq) {t[22]-t[21]:6d} | update+sums+p {t[27]-t[22]:6d} | total {t[27]-t[19]:6d}")
else:
    names = ['assemble', 'PCG attempt / skipped', 'cholesky', 'solve1', 'residual', 'solve2']
    print(f"assemble {t[1]-t[0]:8d}\nPCG (not converged or HM_FORCE_DIRECT_SOLVE) {t[2]-t[1]:8d}")
    print(f"cholesky {t[16]-t[2]:8d}\nsolve1 {t[3]-t[16]:8d}\nresidual {t[4]-t[3]:8d}\nsolve2 {t[5]-t[4]:8d}")
    print('total to end of solve', t[6]-t[0])
    nm = ['publish+barrier', 'diag factor (wave 0)', 'barrier', 'panel solve (wave 0 rows)', 'barrier', 'trailing update + copy-out', 'barrier']
    for i, n in enumerate(nm): print(f'  block column 0: {n:28s} {t[9+i]-t[8+i]:8d} ticks')
