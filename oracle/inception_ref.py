"""CPU oracle (TEST INFRASTRUCTURE ONLY) of the FID feature extractor: the `InceptionV3` wrapper of pytorch-fid 0.3.0
(requirement.txt:151), which the reference's fid_score.py:50,91-148,264-270 instantiates as `InceptionV3([block_idx])`
and feeds `ToTensor()` batches in [0, 1].

**Parity unpinned**: pytorch-fid (and torchvision's Inception3 it is built on) is a third-party dependency that is not
under /root/reference and not installed here; this file restates its published architecture in plain fp32 torch:

* BasicConv2d = Conv2d(bias=False) -> BatchNorm2d(eps=1e-3, running statistics) -> ReLU;
* stem Conv2d_1a_3x3 (3->32, stride 2), 2a (32->32), 2b (32->64, pad 1), MaxPool(3, 2), 3b_1x1 (64->80), 4a_3x3 (80->192), MaxPool(3, 2);
* Mixed_5b/5c/5d = FIDInceptionA(192|256|288, pool_features 32|64|64); Mixed_6a = InceptionB(288);
  Mixed_6b..6e = FIDInceptionC(768, c7 = 128|160|160|192); Mixed_7a = InceptionD(768);
  Mixed_7b = FIDInceptionE_1(1280), Mixed_7c = FIDInceptionE_2(2048); AdaptiveAvgPool2d(1) -> pool3 [B, 2048, 1, 1];
* the "FID" variants differ from torchvision's blocks only in their pooling branch: 3x3 average pooling that EXCLUDES the
  zero padding from the divisor (count_include_pad=False, TensorFlow semantics), and max pooling in Mixed_7c;
* the wrapper resizes to 299 x 299 (bilinear, align_corners=False) and maps [0, 1] -> [-1, 1] (2x - 1);
* state-dict keys are torchvision's (`Mixed_5b.branch1x1.conv.weight`, `.bn.running_mean`, ...), so the published
  `pt_inception-2015-12-05-6726825d.pth` loads unchanged (its `fc.*` entries are ignored: FID stops at pool3).
Known answers that pin the restatement: 21 785 568 convolution + BatchNorm parameters up to pool3 = torchvision's published
27 161 264 for Inception3 minus its AuxLogits head (3 326 696) and fc 2048 -> 1000 (2 049 000); output
shapes per block (64 x 73 x 73 after block 0 for a 299 input, 192 x 35 x 35, 768 x 17 x 17, 2048 x 1 x 1).
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


def _avg3(x):          # TensorFlow-style average pooling: padded zeros are not counted
    return F.avg_pool2d(x, kernel_size=3, stride=1, padding=1, count_include_pad=False)


class InceptionA(nn.Module):
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1 = BasicConv2d(cin, 48, kernel_size=1)
        self.branch5x5_2 = BasicConv2d(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, padding=1)
        self.branch_pool = BasicConv2d(cin, pool_features, kernel_size=1)

    def forward(self, x):
        return torch.cat([self.branch1x1(x), self.branch5x5_2(self.branch5x5_1(x)),
                          self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))), self.branch_pool(_avg3(x))], 1)


class InceptionB(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        return torch.cat([self.branch3x3(x), self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                          F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class InceptionC(nn.Module):
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7_2 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))))
        return torch.cat([self.branch1x1(x), b7, bd, self.branch_pool(_avg3(x))], 1)


class InceptionD(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch3x3_2 = BasicConv2d(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        return torch.cat([self.branch3x3_2(self.branch3x3_1(x)), b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class InceptionE(nn.Module):
    def __init__(self, cin, pool="avg"):
        super().__init__()
        self.pool = pool
        self.branch1x1 = BasicConv2d(cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b3 = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(b3), self.branch3x3_2b(b3)], 1)
        bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(bd), self.branch3x3dbl_3b(bd)], 1)
        bp = _avg3(x) if self.pool == "avg" else F.max_pool2d(x, kernel_size=3, stride=1, padding=1)
        return torch.cat([self.branch1x1(x), b3, bd, self.branch_pool(bp)], 1)


class InceptionV3Ref(nn.Module):
    """pytorch-fid 0.3.0 `InceptionV3(output_blocks, resize_input=True, normalize_input=True)`: returns the list of the
    requested blocks' outputs (fid_score.py:132 takes `model(batch)[0]`)."""
    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}

    def __init__(self, output_blocks: Sequence[int] = (3,), resize_input: bool = True, normalize_input: bool = True):
        super().__init__()
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, "Last possible output block index is 3"
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.Mixed_5b = InceptionA(192, 32)
        self.Mixed_5c = InceptionA(256, 64)
        self.Mixed_5d = InceptionA(288, 64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, 128)
        self.Mixed_6c = InceptionC(768, 160)
        self.Mixed_6d = InceptionC(768, 160)
        self.Mixed_6e = InceptionC(768, 192)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280, "avg")
        self.Mixed_7c = InceptionE(2048, "max")
        self.eval()
        for p in self.parameters():
            p.requires_grad_(False)

    def randomize(self, seed: int = 0):
        """Non-trivial BatchNorm statistics and He-scaled convolutions (tests: there are no published weights on the box)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    fan = m.weight[0].numel()
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                    m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                    m.running_mean.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                    m.running_var.copy_(1.0 + 0.3 * torch.rand(m.bias.shape, generator=g))
        return self

    @torch.no_grad()
    def forward(self, inp):
        out = []
        x = inp
        if self.resize_input:
            x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
        if self.normalize_input:
            x = 2 * x - 1
        blocks = (
            lambda x: F.max_pool2d(self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x))), kernel_size=3, stride=2),
            lambda x: F.max_pool2d(self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x)), kernel_size=3, stride=2),
            lambda x: self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))))))),
            lambda x: F.adaptive_avg_pool2d(self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x))), (1, 1)),
        )
        for idx, blk in enumerate(blocks):
            x = blk(x)
            if idx in self.output_blocks:
                out.append(x)
            if idx == self.last_needed_block:
                break
        return out
