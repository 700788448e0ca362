"""How accurate would a 3-way bf16 split of the fp32 GEMM operands be?

a = a1 + a2 + a3 with bf16 parts; products kept: 3 terms (a1b1, a1b2, a2b1) or 6 terms (+ a1b3, a2b2, a3b1),
accumulated in fp32 like the bf16 MFMA does.  Reference: the fp64 product.  CPU / numpy only.
"""
import numpy as np

def bf16(x):  # round-to-nearest-even to 8 significant bits, result kept in fp32
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)

def split3(x):
    a1 = bf16(x); r = x - a1
    a2 = bf16(r); r = r - a2
    return a1, a2, bf16(r)

rng = np.random.default_rng(0)
M, K, N = 512, 72, 2048
X = rng.standard_normal((M, K)).astype(np.float32)
W = (rng.uniform(-1, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
ref = X.astype(np.float64) @ W.astype(np.float64).T
scale = np.abs(ref).max()
f32 = (X @ W.T).astype(np.float64)
x1, x2, x3 = split3(X); w1, w2, w3 = split3(W)
mm = lambda a, b: (a @ b.T).astype(np.float32)  # fp32 accumulation of exact bf16 products
t3 = mm(x1, w1) + (mm(x1, w2) + mm(x2, w1))
t6 = t3 + ((mm(x1, w3) + mm(x3, w1)) + mm(x2, w2))
for name, v in (("fp32 GEMM", f32), ("bf16 split, 3 terms", t3), ("bf16 split, 6 terms", t6)):
    print(f"{name:22s} max |err| / max |ref| = {np.abs(v - ref).max() / scale:.2e}")
