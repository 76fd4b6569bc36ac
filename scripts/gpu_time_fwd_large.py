"""Is a forward-only decode of ~93k samples per instance (wild_pepper's ray-sample job) as fast per tile as a small one?"""
import sys, time
sys.path.insert(0, '.')
import torch
from hortimapping_amd import synthetic as S, ops
from hortimapping_amd.decoder import DecoderWeights
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
p = S.make_synthetic_decoder(L, seed=1)
dec = DecoderWeights.from_params(p); dec.set_precision('f16x3')
for B, n in ((64, 1024), (64, 8192), (64, 93440), (8, 93440)):
    lat = (0.07 * torch.randn(B, L)).float().cuda()
    pts4 = torch.zeros(B, n, 4, device='cuda'); pts4[..., :3] = 0.04 * torch.randn(B, n, 3, device='cuda')
    nq = torch.full((B,), n, dtype=torch.int32).cuda()
    for _ in range(2): ops.decode_batch(dec, lat, pts4, nq, mode=0)
    torch.cuda.synchronize()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps): ops.decode_batch(dec, lat, pts4, nq, mode=0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    tiles = B * n / 64
    print(f"L={L} B={B} n={n}: {dt*1e3:.3f} ms, {tiles:.0f} tiles, {dt*1e6/ (tiles/256):.1f} us per tile round, {B*n*3671040/dt/1e12:.1f} TFLOP/s algorithmic")
