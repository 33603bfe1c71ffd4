"""Emulates the rounding of the half-piece Winograd forms along x (F(2,3), F(4,3)) against fp64 on VGG-like data: fp32 transform chain,
half-piece split, three products, fp32 accumulation per 16-channel chunk.  python tools/f43_error_model.py (CPU, ~2 min)."""
import numpy as np
rng = np.random.default_rng(0)
def split(v):           # v float32 already scaled -> hi, lo halves (as float32 values)
    hi = v.astype(np.float16).astype(np.float32)
    lo = (v - hi).astype(np.float16).astype(np.float32)
    return hi, lo
def scale_exp(m, target):
    return target - int(np.floor(np.log2(m)))  # 2^k * m in [2^target, 2^(target+1))
def run(F, Cin=256, Cout=32, H=16, W=32, relu_in=True, mean=0.0):
    x = rng.random((H + 2, W + 2, Cin)).astype(np.float32) if mean else np.maximum(rng.standard_normal((H + 2, W + 2, Cin)).astype(np.float32), 0)
    x[0] = 0; x[-1] = 0; x[:, 0] = 0; x[:, -1] = 0
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    # fp64 reference
    ref = np.zeros((H, W, Cout))
    for ky in range(3):
        for kx in range(3):
            ref += x[ky:ky + H, kx:kx + W].astype(np.float64) @ w[:, :, ky, kx].astype(np.float64).T
    if F == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
        xt, wt = 13 - 1, 9 - 1
    else:
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float32)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float32)
        xt, wt = 13 - 4, 9 - 1
    m, a = BT.shape[1] - 2, BT.shape[0]
    kx_ = scale_exp(np.abs(x).max(), xt); kw_ = scale_exp(np.abs(w).max(), wt)
    # weights: U[q] = G g (fp32 arithmetic in the pack kernel; here fp64 then rounded: close enough), scaled, split
    U = np.einsum('qk,oiyk->yqoi', G, w.astype(np.float64)).astype(np.float32) * np.float32(2.0 ** kw_)
    Uh, Ul = split(U)
    out = np.zeros((H, W, Cout), np.float32)
    for p in range(W // m):
        d = x[:, m * p:m * p + a] * np.float32(2.0 ** kx_)        # [H+2, a, Cin] scaled (exact)
        V = np.zeros((H + 2, a, Cin), np.float32)
        for q in range(a):                                     # fp32 fma chain
            acc = np.zeros((H + 2, Cin), np.float32)
            for j in range(a):
                if BT[q, j] != 0: acc = (acc + BT[q, j] * d[:, j]).astype(np.float32)
            V[:, q] = acc
        Vh, Vl = split(V)
        M = np.zeros((a, H, Cout), np.float32)
        for c0 in range(0, Cin, 16):
          for ky in range(3):
            for q in range(a):
                A_h, A_l = Vh[ky:ky + H, q, c0:c0+16].astype(np.float64), Vl[ky:ky + H, q, c0:c0+16].astype(np.float64)
                B_h, B_l = Uh[ky, q][:, c0:c0+16].astype(np.float64).T, Ul[ky, q][:, c0:c0+16].astype(np.float64).T
                for P in (A_l @ B_h, A_h @ B_l, A_h @ B_h):
                    M[q] = (M[q] + P.astype(np.float32)).astype(np.float32)
        M = (M.astype(np.float32) * np.float32(2.0 ** -(kx_ + kw_))).astype(np.float32)    # (fp32 accumulators: rounding of the final value only, an under-estimate)
        for e in range(m):
            acc = np.zeros((H, Cout), np.float32)
            for q in range(a):
                if AT[e, q] != 0: acc = (acc + AT[e, q] * M[q]).astype(np.float32)
            out[:, m * p + e] = acc
    return np.abs(out - ref).max() / np.abs(ref).max()
for mean in (0.0, 2.0):
    for F in (2, 4):
        print("F(%d,3) mean %.0f Cin 256: %.2e   Cin 512: %.2e" % (F, mean, run(F, mean=mean), run(F, Cin=512, H=8, mean=mean)))
