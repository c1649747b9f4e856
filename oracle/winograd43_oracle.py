"""TEST INFRASTRUCTURE ONLY: numpy restatement of the F(4x4, 3x3) Winograd convolution the fp32 plan runs its 3x3 convs in
(csrc/conv_wino4.hip; the module boundary is nn.Conv2d(k=3, padding=1) of lib/network/rtpose_vgg.py:23-35), built from
the exact Toom-Cook matrices of oracle/winograd_tables.py with every intermediate rounded to float32 where the kernel
rounds: input transform in two 1-D passes (along x, then along y), transformed filters rounded once, products summed over
the input channels in float32, output transform in two passes, bias added last.  The order of the float32 additions
inside a matrix-core instruction is not modelled, so the restatement is compared with the kernel - and with the float64
direct sum - by the element-wise bound |err| <= gamma 2^-24 sum |x||w| (tests/test_oracle_cpu.py,
tests/test_wino_numerics_gpu.py), not bit for bit."""
import numpy as np

from .winograd_tables import POINTS_F4_3, toom_cook


def _f(M, dt):
    return np.array([[float(v) for v in row] for row in M], dtype=dt)


def conv3x3_f43(x, w, bias=None):
    """x [cin, H, W] float32, w [cout, cin, 3, 3] float32 -> y [cout, H, W] float32 (zero padding 1)."""
    AT, G, BT = toom_cook(4, 3, POINTS_F4_3)
    BT32, AT32, G64 = _f(BT, np.float32), _f(AT, np.float32), _f(G, np.float64)
    cin, H, W = x.shape
    cout = w.shape[0]
    TY, TX = (H + 3) // 4, (W + 3) // 4
    xp = np.zeros((cin, 4 * TY + 2, 4 * TX + 2), dtype=np.float32)
    xp[:, 1:H + 1, 1:W + 1] = x
    # U = G g G^T in double, one rounding (pack_wino4_kernel)
    U = np.einsum('ak,ockl,bl->ocab', G64, w.astype(np.float64), G64).astype(np.float32)
    y = np.zeros((cout, 4 * TY, 4 * TX), dtype=np.float32)
    for ty in range(TY):
        for tx in range(TX):
            d = xp[:, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]
            t = np.einsum('bx,cyx->cyb', BT32.astype(np.float64), d.astype(np.float64)).astype(np.float32)   # along x
            V = np.einsum('ay,cyb->cab', BT32.astype(np.float64), t.astype(np.float64)).astype(np.float32)   # along y
            M = np.zeros((cout, 6, 6), dtype=np.float32)
            for c in range(cin):
                M = (M + (U[:, c] * V[c][None]).astype(np.float32)).astype(np.float32)
            s = np.einsum('ia,oab->oib', AT32.astype(np.float64), M.astype(np.float64)).astype(np.float32)    # along y
            o = np.einsum('jb,oib->oij', AT32.astype(np.float64), s.astype(np.float64)).astype(np.float32)    # along x
            y[:, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = o
    y = y[:, :H, :W]
    if bias is not None:
        y = (y + bias[:, None, None].astype(np.float32)).astype(np.float32)
    return y
