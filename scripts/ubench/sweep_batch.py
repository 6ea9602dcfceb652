import sys, torch
sys.path.insert(0, '.')
from lc_amd import _lib, synth
lib = _lib.load(); P = _lib.ptr
dev = torch.device('cuda:0')
for B in (768, 896, 960, 1000, 1016, 1023, 1024, 1025, 1032, 1088, 1280, 1536, 2047, 2048, 2049, 3072):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, 64, seed=0).items()}
    st = torch.empty_like(b['start']); tr = torch.empty(B, device=dev); ret = torch.empty(B, device=dev, dtype=torch.int32)
    loss = torch.empty(B, device=dev); du = torch.empty_like(b['pts2d']); ds = torch.empty_like(b['pts2d']); dx = torch.empty_like(b['pts3d'])
    def pnp():
        lib.lc_pnp_lm3_f32(P(b['K']), P(b['pts3d']), P(b['pts2d']), None, P(b['inv_std']), None, None, P(b['start']), P(st), P(tr), P(ret), None, B, 64, 50, 1e-6, 0, 0, None, 0, None)
    def los():
        lib.lc_cov_loss3_fwd_bwd_f32(P(b['K']), P(b['pose']), P(b['pts3d']), P(b['pts2d']), P(b['inv_std']), None, P(b['bbox_3d']), None, B, 64, 32.0, 3.0, 4.0, 0, P(loss), P(du), P(ds), P(dx), None, None, 0, None)
    out = []
    for fn in (pnp, los):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(B, 'pnp %.1f us  loss %.1f us' % tuple(out))
