"""The zlmo / zycbv backbone shape in plain PyTorch-ROCm (MIOpen / hipBLASLt own every GEMM; nothing here is part of the hot path
this repo re-implements -- SURVEY.md section 2 row 9): a 34-layer residual encoder held at output stride 8 by dilation, an atrous
spatial pyramid on top, and a two-stage up-sampling decoder with skip concatenations that ends in 128x128 maps for a 256x256 crop.

Structure after `model/zebra_resnet.py:171-255` (stages 4 and 5 keep stride 1 and dilate by 2 and 4) and
`model/zebra_DeepLabV3.py:28-169` (1x1 + three dilated 3x3 branches at rates 6 / 12 / 18 + image pooling, fused by a 1x1; transposed
convolution + two 3x3 per up-sampling stage; the 64x64 and 128x128 encoder features concatenated before / after the second stage).
Random-init weights: the example measures the step, it does not train a model.

    crop (B,3,256,256) -> stem /2 -> x128 (w) -> pool /2 + stage1 -> x64 (w) -> stage2 /2 -> x32 (2w)
      -> stage3 dil 2 (4w) -> stage4 dil 4 (8w) -> pyramid (256) -> up x2 (256) ++ x64 -> up x2 (256) ++ x128 = feature (256 + w)
      -> 1x1 -> (B,C,128,128)
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _cbr(cin, cout, k, dilation=1, stride=1, bias=False):
    pad = dilation * (k // 2)
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, pad, dilation, bias=bias), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DilatedBlock(nn.Module):
    """Two 3x3 convolutions at one dilation rate + identity / projected shortcut."""

    def __init__(self, cin, cout, stride=1, dilation=1):
        super().__init__()
        self.a = _cbr(cin, cout, 3, dilation, stride)
        self.b = nn.Sequential(nn.Conv2d(cout, cout, 3, 1, dilation, dilation, bias=False), nn.BatchNorm2d(cout))
        self.short = None
        if stride != 1 or cin != cout:
            self.short = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        return F.relu(self.b(self.a(x)) + (x if self.short is None else self.short(x)))


def _stage(cin, cout, blocks, stride=1, dilation=1):
    return nn.Sequential(*[DilatedBlock(cin if i == 0 else cout, cout, stride if i == 0 else 1, dilation) for i in range(blocks)])


class Pyramid(nn.Module):
    """Atrous spatial pyramid: five 256-channel views of the stride-8 feature (point-wise, three dilated, image-level), fused by a 1x1."""

    def __init__(self, cin, mid=256, rates=(6, 12, 18)):
        super().__init__()
        self.point = _cbr(cin, mid, 1, bias=True)
        self.atrous = nn.ModuleList([_cbr(cin, mid, 3, r, bias=True) for r in rates])
        self.image = _cbr(cin, mid, 1, bias=True)
        self.fuse = _cbr(mid * (2 + len(rates)), mid, 1, bias=True)

    def forward(self, x):
        views = [self.point(x)] + [m(x) for m in self.atrous]
        # the image-level view is constant over the map: bilinear interpolation of a 1x1 map = a broadcast
        views.append(self.image(x.mean((-2, -1), keepdim=True)).expand(-1, -1, *x.shape[-2:]))
        return self.fuse(torch.cat(views, 1))


def _up(cin, cout=256):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 3, 2, 1, output_padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         *_cbr(cout, cout, 3), *_cbr(cout, cout, 3))


class OS8Trunk(nn.Module):
    """-> (raw maps (B, out_channels, H/2, W/2), feature (B, 256 + width, H/2, W/2)); `feature_dim` as `zebra_DeepLabV3.get_network` sets it."""

    def __init__(self, out_channels, width=64, layers=(3, 4, 6, 3)):
        super().__init__()
        w = width
        self.stem = _cbr(3, w, 7, stride=2)
        self.stage1 = nn.Sequential(nn.MaxPool2d(3, 2, 1), _stage(w, w, layers[0]))
        self.stage2 = _stage(w, 2 * w, layers[1], stride=2)
        self.stage3 = _stage(2 * w, 4 * w, layers[2], dilation=2)
        self.stage4 = _stage(4 * w, 8 * w, layers[3], dilation=4)
        self.pyramid = Pyramid(8 * w)
        self.up1 = _up(256)
        self.up2 = _up(256 + w)
        self.feature_dim = 256 + w
        self.head = nn.Conv2d(self.feature_dim, out_channels, 1)

    def forward(self, rgb):
        x128 = self.stem(rgb)
        x64 = self.stage1(x128)
        top = self.pyramid(self.stage4(self.stage3(self.stage2(x64))))
        feature = torch.cat((self.up2(torch.cat((self.up1(top), x64), 1)), x128), 1)
        return self.head(feature), feature
