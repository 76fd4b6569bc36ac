import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from hortimapping_amd import synthetic as S
from hortimapping_amd.decoder import DecoderWeights
from hortimapping_amd import optimizer as HO
from oracle import hm_oracle as O
from tests.golden_util import load, list_golden, decoder_params, cfg_from_golden, render_data_from_golden, relmax

decs = {}
def get(name):
    name = str(name)
    if name not in decs:
        p = decoder_params(name)
        decs[name] = (DecoderWeights.from_params(p), O.fold_decoder(p))
    return decs[name]

def inst_from_golden(g, z0=None):
    return HO.Instance(torch.from_numpy(g['latent0'] if z0 is None else z0), torch.from_numpy(g['T_ow0']),
                       torch.from_numpy(g['points_w']), render_data_from_golden(g), float(g['cube_radius']), False)

# one-iteration parity (G8)
for name in ('pepper32', 'pepper256'):
    g = load(f'g8_one_iter_{name}')
    dec, od = get(name)
    cfg = cfg_from_golden(g)
    L = dec.latent_dim; P = 7
    inst = inst_from_golden(g, g['z0'])
    dbg = {}
    res = HO.optimize_batch(dec, cfg, [inst], shape_only=False, debug=dbg)[0]
    A = dbg['A'][0].cpu().numpy(); b = dbg['b'][0].cpu().numpy(); d = dbg['delta'][0].cpu().numpy()
    E = L + P
    A = np.tril(A[:E, :E]); A = A + A.T - np.diag(np.diag(A))
    perm = list(range(L, L + P)) + list(range(L))      # reference order [pose | code]
    Ar = A[np.ix_(perm, perm)]; br = b[perm]; dr = d[perm]
    print(name, 'joint: counts', dbg['counts'][0].cpu().numpy(), 'H', relmax(Ar, g['H_free']), 'b', relmax(br, g['b_free']),
          'delta', relmax(dr, g['delta_free']), 'z', relmax(res.latent, g['z_free']), 'T', relmax(res.T_ow, g['T_free']), 'it', res.iter_count, 'st', res.status)
    # fp64 oracle delta for reference
    tr = []
    O.shape_pose_joint_opt(od.to(torch.float64), cfg, torch.from_numpy(g['z0']), torch.from_numpy(g['T_ow0']), render_data_from_golden(g), torch.from_numpy(g['points_w']), float(g['cube_radius']), trace=tr)
    print('   vs fp64 oracle: delta gpu', relmax(dr, tr[0].delta), ' ref-golden', relmax(g['delta_free'], tr[0].delta), 'H gpu', relmax(Ar, tr[0].H), 'H ref', relmax(g['H_free'], tr[0].H))
    dbg = {}
    res = HO.optimize_batch(dec, cfg, [inst], shape_only=True, debug=dbg)[0]
    A = dbg['A'][0].cpu().numpy()[:L, :L]; A = np.tril(A); A = A + A.T - np.diag(np.diag(A))
    print(name, 'sdf: H', relmax(A, g['H_sdf']), 'b', relmax(dbg['b'][0].cpu().numpy()[:L], g['b_sdf']), 'delta', relmax(dbg['delta'][0].cpu().numpy()[:L], g['delta_sdf']), 'z', relmax(res.latent, g['z_sdf']))

# trajectories (G9), batched per decoder
for name in list_golden('g9_traj_'):
    g = load(name)
    if 'lin8_bias_shift' in g.files:
        continue                     # shifted-decoder cases: tests/test_gpu_parity.py::test_trajectories_vs_golden
    dec, od = get(g['decoder'])
    cfg = cfg_from_golden(g)
    inst = inst_from_golden(g); inst.pose_known = bool(g['pose_known'])
    res = HO.optimize_batch(dec, cfg, [inst], shape_only=(str(g['kind']) == 'sdf'))[0]
    zr = relmax(res.latent, g['z_out']) if np.abs(g['z_out']).max() > 0 else float(res.latent.abs().max())
    print(f"{name[8:]:22s} it {res.iter_count:3d}/{int(g['iter_count']):3d} st {res.status:2d} z {zr:.2e} T {relmax(res.T_ow, g['T_out']):.2e}")
