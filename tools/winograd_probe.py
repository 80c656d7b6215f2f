"""Go / no-go probe for a Winograd F(2x2, 3x3) form of the shared 3x3 spike convolution.
CPU only.  Transformed spikes B^T d B are small integers (exact in bf16); transformed weights G g G^T are formed in fp64 and
cut to `planes` bf16 planes (4 planes = 32 bits); the 16 element-wise GEMMs over the channels accumulate in fp32; the output
transform A^T M A runs in fp32.  Measured against an fp64 convolution: rms / max error of the input currents next to the
direct fp32 convolution's, and the LIF spikes that flip against the oracle's spike train over T steps."""
import sys
import os
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import snn_oracle as OR

torch.set_num_threads(8)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def to_planes(v64, planes):
    """fp64 -> sum of `planes` bf16 values (round to nearest each), returned as fp32 tensors"""
    out, r = [], v64.clone()
    for _ in range(planes):
        p = r.to(torch.float32).to(torch.bfloat16)
        out.append(p.to(torch.float32))
        r = r - p.to(torch.float64)
    return out


def winograd_conv(z, w, planes):
    """z [N,C,H,W] in {0,1} (fp32), w [Co,C,3,3] fp32 -> [N,Co,H,W] fp32 (pad 1); H, W even"""
    N, C, H, W = z.shape
    Co = w.shape[0]
    V = torch.einsum("ia,ocab,jb->ijoc", G, w.double(), G)                 # [4,4,Co,C] fp64
    Vp = to_planes(V, planes)
    zp = F.pad(z, (1, 1, 1, 1))
    tiles = zp.unfold(2, 4, 2).unfold(3, 4, 2)                             # [N,C,H/2,W/2,4,4]
    U = torch.einsum("ia,nchwab,jb->ijnhwc", Bt.float(), tiles, Bt.float())   # exact small integers
    U = U.reshape(4, 4, -1, C)
    M = torch.zeros(4, 4, U.shape[2], Co, dtype=torch.float32)
    for p in reversed(Vp):                                                 # small planes first, fp32 accumulation (one chain per plane)
        M = M + torch.einsum("ijtc,ijoc->ijto", U, p)
    Y = torch.einsum("ai,ijto,bj->tabo", At.float(), M, At.float())        # fp32 output transform
    Y = Y.reshape(N, H // 2, W // 2, 2, 2, Co).permute(0, 5, 1, 3, 2, 4).reshape(N, Co, H, W)
    return Y


def main():
    T, C = 8, 256
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(2, C, 48, 96, generator=g) * 1.5                    # firing rates close to the backbone-fed pyramid's
    w = torch.randn(C, C, 3, 3, generator=g) * 0.01
    z = OR.encoder_spikes(feat, T)                                         # [T,N,C,H,W]
    cur64 = torch.stack([F.conv2d(z[t].double(), w.double(), padding=1) for t in range(T)])
    cur32 = torch.stack([F.conv2d(z[t], w, padding=1) for t in range(T)])
    spk_ref, _, vdec = OR.lif_scan_from_currents(cur32)
    print("encoder rate %.3f, shared-LIF rate %.3f, neuron-steps %.3g" % (float(z.mean()), float(spk_ref.mean()), spk_ref.numel()))
    e = (cur32.double() - cur64)
    print("direct fp32 conv      : rms err %.3g  max %.3g" % (float(e.pow(2).mean().sqrt()), float(e.abs().max())))
    # a second fp32 summation order of the direct convolution (what the HIP kernel's order amounts to): unfold + matmul
    cols = torch.stack([F.unfold(z[t], 3, padding=1) for t in range(T)])   # [T,N,C*9,HW]
    alt = torch.einsum("ok,tnkp->tnop", w.reshape(C, -1), cols).reshape(cur32.shape)
    e = (alt.double() - cur64)
    spk_alt, _, _ = OR.lif_scan_from_currents(alt)
    print("direct, other order   : rms err %.3g  max %.3g   flipped neuron trains %d" % (
        float(e.pow(2).mean().sqrt()), float(e.abs().max()), int((spk_alt != spk_ref).any(0).sum())))
    for planes in (3, 4, 5):
        wc = torch.stack([winograd_conv(z[t], w, planes) for t in range(T)])
        e = (wc.double() - cur64)
        spk_w, _, _ = OR.lif_scan_from_currents(wc)
        flipped = (spk_w != spk_ref).any(0)
        pos = flipped.any(dim=1)                                           # positions (n, y, x) holding a flip
        print("winograd, %d bf16 planes: rms err %.3g  max %.3g   flipped neuron trains %d in %d positions of %d" % (
            planes, float(e.pow(2).mean().sqrt()), float(e.abs().max()), int(flipped.sum()), int(pos.sum()), pos.numel()))
    # MFMA work model (bf16 16x16x32 plane-MACs per output element)
    direct = 9 * C * 3
    for planes in (3, 4, 5):
        wino = 16 * C * planes / 4
        print("plane-MACs per output: direct %d, winograd(%d planes) %d  -> x%.2f fewer" % (direct, planes, wino, direct / wino))


if __name__ == "__main__":
    main()
