#!/usr/bin/env python3
"""Gate for F(4x4,3x3) (VERDICT r04 "next" #8): fp32 emulation on torch-CPU of the three passes of a 3x3 / stride-1 / SAME
layer in the F(4x4,3x3) Winograd domain against fp64 direct convolution, on D l3 / l5 / l7 shapes (batch reduced: the error
does not depend on N for forward / dgrad; for wgrad the tile count is kept at the CIFAR step's 2B = 128).  F(2x2,3x3) - what
ships - beside it.  Prints element-wise error / max|ref| and L2 error per pass.  Build a kernel only if element-wise <= 2e-5.
    python tools/wino43_gate.py [--points lavin|small]"""
import sys
import numpy as np
import torch

torch.manual_seed(0)
torch.set_num_threads(8)


def mats(m, pts):
    """Cook-Toom matrices (A^T [m x a], G [a x 3], B^T [a x a]) for F(m, 3) with interpolation points pts (+ infinity)"""
    import sympy as sp
    a = m + 2
    pts = [sp.Rational(p) for p in pts]
    assert len(pts) == a - 1
    x = sp.symbols('x')
    # following Lavin / wincnn
    def At():
        return sp.Matrix(a - 1, m, lambda i, j: pts[i] ** j).T.row_join(sp.Matrix(m, 1, lambda i, j: 1 if i == m - 1 else 0))
    f = [sp.prod([pts[i] - pts[k] for k in range(a - 1) if k != i]) for i in range(a - 1)]
    G = sp.Matrix(a - 1, 3, lambda i, j: pts[i] ** j / f[i]).col_join(sp.Matrix(1, 3, lambda i, j: 1 if j == 2 else 0))
    M = sp.prod([x - p for p in pts])
    Bt = sp.zeros(a, a)
    for i in range(a - 1):
        q = sp.Poly(sp.quo(M, x - pts[i]), x).all_coeffs()[::-1]
        for j, c in enumerate(q):
            Bt[i, j] = c
    qm = sp.Poly(M, x).all_coeffs()[::-1]
    for j, c in enumerate(qm):
        Bt[a - 1, j] = c
    AT = At()
    return (np.array(AT.tolist(), dtype=np.float64), np.array(G.tolist(), dtype=np.float64), np.array(Bt.tolist(), dtype=np.float64))


def check_mats(AT, G, BT):
    m, a = AT.shape
    rs = np.random.RandomState(0)
    d, g = rs.randn(a), rs.randn(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(3)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-9), (y, ref)


def tiles(x, m):
    """x [N,C,H,W] -> overlapping (m+2)x(m+2) patches of the zero-padded input: [N,C,TH,TW,a,a]"""
    a = m + 2
    xp = torch.nn.functional.pad(x, (1, 1 + (-x.shape[3]) % m, 1, 1 + (-x.shape[2]) % m))
    return xp.unfold(2, a, m).unfold(3, a, m)


def wino_fwd(x, w, AT, G, BT, dt):
    """x [N,C,H,W], w [K,C,3,3] -> y [N,K,H,W]; every stage in dtype dt"""
    m = AT.shape[0]
    AT, G, BT = (torch.tensor(t, dtype=dt) for t in (AT, G, BT))
    N, C, H, W = x.shape
    d = tiles(x.to(dt), m)                                              # N C TH TW a a
    V = torch.einsum('ia,nctuab,jb->ijnctu', BT, d, BT)                # a a N C TH TW
    U = torch.einsum('ia,kcab,jb->ijkc', G, w.to(dt), G)               # a a K C
    M = torch.einsum('ijnctu,ijkc->ijnktu', V, U)
    Y = torch.einsum('pi,ijnktu,qj->nktpuq', AT, M, AT)                # N K TH m TW m
    TH, TW = Y.shape[2], Y.shape[4]
    return Y.reshape(N, -1, TH * m, TW * m)[:, :, :H, :W]


def wino_wgrad(x, dy, AT, G, BT, dt):
    """dW [K,C,3,3] = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G"""
    m = AT.shape[0]
    AT, G, BT = (torch.tensor(t, dtype=dt) for t in (AT, G, BT))
    N, C, H, W = x.shape
    K = dy.shape[1]
    d = tiles(x.to(dt), m)
    V = torch.einsum('ia,nctuab,jb->ijnctu', BT, d, BT)
    TH, TW = d.shape[2], d.shape[3]
    dyp = torch.nn.functional.pad(dy.to(dt), (0, TW * m - W, 0, TH * m - H)).reshape(N, K, TH, m, TW, m)
    E = torch.einsum('pi,nktpuq,qj->ijnktu', AT, dyp, AT)              # A dY A^T: a a N K TH TW
    S = torch.einsum('ijnctu,ijnktu->ijkc', V, E)
    return torch.einsum('ia,ijkc,jb->kcab', G, S, G)


def main():
    sets = {'F(2x2,3x3)': (2, [0, 1, -1]), 'F(4x4,3x3) lavin 0,1,-1,2,-2': (4, [0, 1, -1, 2, -2]),
            'F(4x4,3x3) 0,1,-1,1/2,-1/2': (4, [0, 1, -1, '1/2', '-1/2']), 'F(4x4,3x3) 0,1,-1,1/2,-2': (4, [0, 1, -1, '1/2', -2])}
    shapes = [('D l3', 16, 128, 128, 16), ('D l5', 16, 256, 256, 8), ('D l7', 32, 512, 512, 4)]
    for name, (m, pts) in sets.items():
        AT, G, BT = mats(m, pts)
        check_mats(AT, G, BT)
        print('== %s' % name)
        for lname, N, C, K, H in shapes:
            x = torch.randn(N, C, H, H, dtype=torch.float64)
            x = torch.where(x > 0, x, 0.1 * x)                          # lrelu-shaped activations, as D's layers see
            w = torch.randn(K, C, 3, 3, dtype=torch.float64) / np.sqrt(9 * C)
            dy = torch.randn(N, K, H, H, dtype=torch.float64)
            ref_y = torch.nn.functional.conv2d(x, w, padding=1)
            ref_dx = torch.nn.functional.conv_transpose2d(dy, w, padding=1)
            ref_dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1)
            out = []
            for what, got, ref in (('fwd', wino_fwd(x.float(), w.float(), AT, G, BT, torch.float32), ref_y),
                                   ('dgrad', wino_fwd(dy.float(), w.float().flip(2, 3).transpose(0, 1), AT, G, BT, torch.float32), ref_dx),
                                   ('wgrad', wino_wgrad(x.float(), dy.float(), AT, G, BT, torch.float32), ref_dw)):
                e = (got.double() - ref).abs()
                out.append('%s max %.2e L2 %.2e' % (what, float(e.max() / ref.abs().max()), float(e.norm() / ref.norm())))
            # the direct fp32 product's own error, for scale
            e0 = (torch.nn.functional.conv2d(x.float(), w.float(), padding=1).double() - ref_y).abs()
            print('  %s N=%d C=%d K=%d H=%d: %s | direct fp32 fwd max %.2e' % (lname, N, C, K, H, '; '.join(out), float(e0.max() / ref_y.abs().max())))


if __name__ == '__main__':
    main()
