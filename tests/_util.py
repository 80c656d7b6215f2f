"""helpers shared by the parity tests: spike bit-plane <-> dense conversion (little-endian bits:
bit b of word w = channel 32*w + b, see include/snn_hip.h)"""
import numpy as np
import torch


def planes_to_dense(planes: torch.Tensor, K: int) -> np.ndarray:
    """int32 [T, M, Kw] -> float32 [T, M, K]"""
    a = planes.detach().cpu().numpy().view(np.uint32)
    T, M, Kw = a.shape
    bits = np.unpackbits(a.view(np.uint8).reshape(T, M, Kw * 4), axis=2, bitorder="little")
    return bits[:, :, :K].astype(np.float32)


def dense_to_planes(z: np.ndarray) -> torch.Tensor:
    """{0,1} array [T, M, K] -> int32 [T, M, Kw] (CPU tensor)"""
    T, M, K = z.shape
    Kw = (K + 31) // 32
    pad = np.zeros((T, M, Kw * 32), dtype=np.uint8)
    pad[:, :, :K] = z.astype(np.uint8)
    words = np.packbits(pad, axis=2, bitorder="little").view(np.uint32).reshape(T, M, Kw)
    return torch.from_numpy(words.view(np.int32).copy())


def nchw_to_rows(z: torch.Tensor) -> np.ndarray:
    """[T, N, C, H, W] -> [T, N*H*W, C] (row = (n*H + y)*W + x)"""
    T, N, C, H, W = z.shape
    return z.permute(0, 1, 3, 4, 2).reshape(T, N * H * W, C).numpy()


def flip_budget(positions: int, channels: int, steps: int) -> float:
    """How many positions may hold a hidden spike that differs from the oracle's.  Two fp32 summation orders
    of the same 3x3 convolution disagree on ~7e-8 of the neuron-steps (SURVEY.md §7 risk 1 measured 5 of 7.5e7
    between two oneDNN layouts); a position has channels*steps of them.  Budget = 3.5x that rate + 2."""
    return 2 + 2.5e-7 * channels * steps * positions
