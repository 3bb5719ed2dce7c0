"""The arithmetic conv3x3_wino runs, restated in numpy: Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output tile (Lavin & Gray
2016) with the matrices of cnn.hip (pack_wino forms G g G^T in f64 and rounds once; the kernel's transforms are f32
additions). Pins the matrices and the claim DESIGN.md makes about the numerics: on post-activation data the f32 Winograd
result is as close to an f64 convolution as a direct f32 convolution is."""
import numpy as np
import torch

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def winograd_conv(x, w, dtype):
    """x [Cin,H,W], w [Cout,Cin,3,3], zero padding 1, stride 1; transforms and accumulation in `dtype`."""
    cin, H, W = x.shape
    U = (G @ w.astype(np.float64) @ G.T).astype(dtype)                       # [Cout,Cin,4,4], rounded once
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(dtype)
    T = H // 2
    d = np.lib.stride_tricks.sliding_window_view(xp, (4, 4), axis=(1, 2))[:, ::2, ::2]      # [Cin,T,T,4,4]
    bt = BT.astype(dtype)
    V = np.einsum('ia,ctuab,jb->ctuij', bt, d, bt).astype(dtype)
    M = np.einsum('kcij,ctuij->ktuij', U, V).astype(dtype)
    at = AT.astype(dtype)
    Y = np.einsum('ai,ktuij,bj->ktuab', at, M, at).astype(dtype)
    return Y.transpose(0, 1, 3, 2, 4).reshape(w.shape[0], H, W)


def test_winograd_f2x2_3x3_is_the_convolution():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((6, 16, 16))
    w = rng.standard_normal((5, 6, 3, 3))
    ref = torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()
    np.testing.assert_allclose(winograd_conv(x, w, np.float64), ref, rtol=0, atol=1e-12)


def test_f32_winograd_is_as_close_to_f64_as_a_direct_f32_convolution():
    rng = np.random.default_rng(1)
    cin, cout, H = 80, 80, 32
    x = np.abs(rng.standard_normal((cin, H, H))).astype(np.float32)           # post-LeakyReLU-like activations
    w = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), padding=1)[0].numpy()
    direct = torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()
    wino = winograd_conv(x, w, np.float32)
    e_direct, e_wino = np.abs(direct - ref).max(), np.abs(wino - ref).max()
    r_direct, r_wino = np.sqrt(np.mean((direct - ref) ** 2)), np.sqrt(np.mean((wino - ref) ** 2))
    assert e_wino < 2 * e_direct + 1e-6 and r_wino < 1.5 * r_direct + 1e-7, (e_direct, e_wino, r_direct, r_wino)
    assert e_wino < 1e-5 * max(1.0, np.abs(ref).max())
