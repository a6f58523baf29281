"""CPU experiment for the round-2 plan: does a Winograd F(2x2, 3x3) form of the stride-1 3x3
convolutions stay inside the error budget when every product is a split-binary16 ("f16x3") product with float32
accumulation, as the MFMA path computes it?  Emulated in numpy; compared with a float64 direct convolution.

  python tools/probe/winograd_numerics.py
"""
import numpy as np


def split16_rtz(x):
    """x (float32) -> (hi, lo) float32 arrays holding binary16 values: hi = rtz16(x), lo = rtz16(x - hi)."""
    def rtz16(v):
        h = v.astype(np.float16)                      # round to nearest ...
        hf = h.astype(np.float32)
        over = np.abs(hf) > np.abs(v)                 # ... then step back towards zero where it rounded away
        h2 = np.nextafter(h, np.float16(0), dtype=np.float16)
        return np.where(over, h2, h).astype(np.float32)
    hi = rtz16(x)
    lo = rtz16((x - hi).astype(np.float32))
    return hi, lo


def mm3(a, b):
    """f16x3 product sum over the last axis of a [.., K] and first of b [K, ..], float32 accumulation."""
    ah, al = split16_rtz(a.astype(np.float32))
    bh, bl = split16_rtz(b.astype(np.float32))
    return (ah @ bh + ah @ bl + al @ bh).astype(np.float32)


def direct(x, w, mm):
    """x [H, W, C], w [Co, C, 3, 3], pad 1 -> [H, W, Co] via im2col and `mm`."""
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C), x.dtype)
    xp[1:-1, 1:-1] = x
    cols = np.stack([xp[kh:kh + H, kw:kw + W] for kh in range(3) for kw in range(3)], 2).reshape(H * W, 9 * C)
    wm = w.transpose(2, 3, 1, 0).reshape(9 * C, -1)
    return mm(cols, wm).reshape(H, W, -1)


BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def winograd(x, w, mm):
    H, W, C = x.shape
    Co = w.shape[0]
    xp = np.zeros((H + 2, W + 2, C), np.float32)
    xp[1:-1, 1:-1] = x
    U = np.einsum("ij,ocjk,lk->iloc", G, w.astype(np.float64), G)        # [4,4,Co,C], offline in float64
    th, tw = H // 2, W // 2
    tiles = np.stack([xp[2 * i:2 * i + 4, 2 * j:2 * j + 4] for i in range(th) for j in range(tw)], 0)  # [T,4,4,C]
    V = np.einsum("ij,tjkc,lk->iltc", BT.astype(np.float32), tiles, BT.astype(np.float32)).astype(np.float32)
    M = np.zeros((4, 4, th * tw, Co), np.float32)
    for a in range(4):
        for b in range(4):
            M[a, b] = mm(V[a, b], U[a, b].T.astype(np.float32))
    Y = np.einsum("ij,jktc,lk->tilc", AT.astype(np.float32), M, AT.astype(np.float32)).astype(np.float32)   # [T,2,2,Co]
    return Y.reshape(th, tw, 2, 2, Co).transpose(0, 2, 1, 3, 4).reshape(H, W, Co)


def main():
    rng = np.random.RandomState(0)
    for C, Co in ((64, 256), (256, 256), (512, 512)):
        H = W = 16
        x = np.maximum(rng.randn(H, W, C), 0).astype(np.float32)          # post-ReLU activations
        w = (rng.randn(Co, C, 3, 3) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
        ref = direct(x.astype(np.float64), w.astype(np.float64), lambda a, b: a @ b)
        scale = np.abs(ref).max()
        rows = []
        for name, fn, mm in (("direct  f32 ", direct, lambda a, b: (a.astype(np.float32) @ b.astype(np.float32))),
                             ("direct  f16x3", direct, mm3),
                             ("winograd f32 ", winograd, lambda a, b: (a.astype(np.float32) @ b.astype(np.float32))),
                             ("winograd f16x3", winograd, mm3)):
            y = fn(x, w, mm)
            rows.append("%s %.2e" % (name, np.abs(y - ref).max() / scale))
        print("Cin %3d Cout %3d  max|err| / max|y|:  " % (C, Co) + "   ".join(rows))


if __name__ == "__main__":
    main()
