import sys, torch
sys.path.insert(0, '.')
which = sys.argv[1]
from lc_amd import synth
dev = torch.device('cuda:0')
if which == 'head':
    from lc_amd.ptnet import spatial_softargmax_2d_std
    lg = synth.make_head_logits(2, 4, 64, 64).to(dev)
    m, s = spatial_softargmax_2d_std(lg); torch.cuda.synchronize(); print('head', m[0,0], s[0,0])
elif which == 'pnp':
    from lc_amd.pnp import pnp_ceres
    b = {k: v.to(dev) for k, v in synth.make_batch(4, 64, seed=0).items()}
    st, tr, ret = pnp_ceres.solve_device(b['K'], b['pts3d'], b['pts2d'], b['inv_std'], b['start']); torch.cuda.synchronize()
    print('pnp', ret, tr, st[0], b['pose'][0])
elif which == 'loss':
    from lc_amd.cov_mixed import loss_cov_mixed_fused
    b = {k: v.to(dev) for k, v in synth.make_batch(4, 64, seed=0).items()}
    out = loss_cov_mixed_fused(b['K'], b['pose'], b['pts3d'], b['pts2d'], b['inv_std'], None, b['bbox_3d']); torch.cuda.synchronize()
    print('loss', out[0])
