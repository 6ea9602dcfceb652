"""CPU oracle of the ZebraPose code decode (TEST INFRASTRUCTURE ONLY): plain-torch restatement of floatbits.py:130-160
(training decode with ground-truth bits), :194-223 (inference Gray decode) and the /(max/2)-1 normalisation (:108-118,
:162-180).  Pinned by tests/golden/bits_*.npz generated from the reference's floatbits module."""
import torch


def decode_with_gt_axis(mod_logits, gt_raw_bits, gt_msk, black_factor):
    """(*,N) logits, (*,N) bool bits, (*) bool mask -> (*) value."""
    gt = gt_raw_bits.to(torch.bool).clone()
    msk = torch.ones_like(mod_logits)
    msk[..., 1:] = torch.where(gt[..., :-1], -torch.ones_like(msk[..., 1:]), msk[..., 1:])
    msk[..., 0:2] = msk[..., 0:2] * black_factor
    logits = mod_logits * msk
    N = logits.shape[-1]
    w = 2.0 ** torch.arange(N - 1, -1, -1, dtype=logits.dtype)
    with torch.no_grad():
        pred = logits > 0
        out_vals = (pred * w).sum(-1)
        err = pred ^ gt
        err[..., -1] = True
        idx = torch.argmax(err.to(torch.uint8), dim=-1, keepdim=True)
        gt_wo = gt.scatter(-1, idx, False)
    correct = (gt_wo * w).sum(-1)
    in_vals = correct + torch.gather(logits, -1, idx).squeeze(-1).sigmoid() * w[idx.squeeze(-1)]
    return torch.where(gt_msk, in_vals, out_vals)


def nn_logits2noc_with_gt(logits, gt_raw_bits, bits, gt_msk, black=True):
    """(B,C,H,W) -> (B,H,W,3)."""
    lg = logits.permute(0, 2, 3, 1)
    gb = gt_raw_bits.permute(0, 2, 3, 1)
    out, c0 = [], 0
    for n in bits:
        v = decode_with_gt_axis(lg[..., c0:c0 + n], gb[..., c0:c0 + n], gt_msk.to(torch.bool), -1 if black else 1)
        out.append(v / ((2 ** n - 1) * 0.5) - 1)
        c0 += n
    return torch.stack(out, -1)


@torch.no_grad()
def nn_logits2noc(logits, bits, black=True):
    lg = logits.permute(0, 2, 3, 1)
    out, c0 = [], 0
    for n in bits:
        l = lg[..., c0:c0 + n]
        b = l > 0
        if black:
            b = b.clone()
            b[..., 0:2] = ~b[..., 0:2]
        code = (b.long() * (2 ** torch.arange(n - 1, -1, -1))).sum(-1)
        v = code.clone()
        sh = 1
        while sh < 32:
            v = v ^ (v >> sh)
            sh <<= 1
        lsb = 1 - (v & 2)
        val = (v & -2).to(l.dtype) + (l[..., -1] * lsb).sigmoid()
        out.append(val / ((2 ** n - 1) * 0.5) - 1)
        c0 += n
    return torch.stack(out, -1)
