import itertools, numpy as np
F = np.float32
d = np.load("/tmp/rz.npz"); z = d["z"]; m = d["m"]; clip = d["clip"]
H, W = z.shape
sx = F(W / 2)
xw = clip[:, 0] * sx + sx; yw = clip[:, 1] * sx + sx; zw = clip[:, 2] * F(0.5) + F(0.5)
def fma(a, b, c): return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)
Xf = np.broadcast_to(np.arange(W, dtype=F)[None, :], (H, W)); Yf = np.broadcast_to(np.arange(H, dtype=F)[:, None], (H, W))
for perm in itertools.permutations(range(3)):
    x, y, zz = xw[list(perm)], yw[list(perm)], zw[list(perm)]
    dx01, dy01, dz01 = x[0] - x[1], y[0] - y[1], zz[0] - zz[1]
    dx20, dy20, dz20 = x[2] - x[0], y[2] - y[0], zz[2] - zz[0]
    ooa = F(1.0) / (dx01 * dy20 - dx20 * dy01)
    dzdx = dz01 * (dy20 * ooa) - dz20 * (dy01 * ooa)
    dzdy = dz20 * (dx01 * ooa) - dz01 * (dx20 * ooa)
    for a0name, a0 in (("plain", zz[0] - (dzdx * (x[0] - F(0.5)) + dzdy * (y[0] - F(0.5)))), ("fma", zz[0] - fma(dzdy, y[0] - F(0.5), dzdx * (x[0] - F(0.5)))), ("fma2", zz[0] - fma(dzdx, x[0] - F(0.5), dzdy * (y[0] - F(0.5))))):
        for ename, zm in (("plain", (a0 + dzdx * Xf) + dzdy * Yf), ("fma_xy", fma(dzdy, Yf, fma(dzdx, Xf, a0))), ("fma_yx", fma(dzdx, Xf, fma(dzdy, Yf, a0)))):
            bad = int((zm.view(np.uint32)[m] != z.view(np.uint32)[m]).sum())
            mx = int(np.abs(zm.view(np.int32)[m].astype(np.int64) - z.view(np.int32)[m]).max())
            print(perm, a0name, ename, bad, mx)
