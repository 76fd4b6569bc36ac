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
names = ['assemble', 'cholesky', 'solve1', 'residual', 'solve2']
for i, n in enumerate(names): print(f"{n:10s} {t[i+1]-t[i]:8d} ticks")
print('total', t[5]-t[0])
nm = ['publish+barrier', 'diag factor (wave 0)', 'barrier', 'panel solve (wave 0 rows)', 'barrier', 'trailing update + copy-out', 'barrier']
for i, n in enumerate(nm): print(f'  block column 0: {n:28s} {t[9+i]-t[8+i]:8d} ticks')
