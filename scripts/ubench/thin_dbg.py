import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layoutdit_amd import ops
dev = 'cuda:0'
os.environ["LDIT_GEMM_TILE"] = "4"
M = N = 32
for K in (32, 64):
    torch.manual_seed(0)
    w = torch.arange(N * K, device=dev, dtype=torch.float32).reshape(N, K) + 1.0   # w[n][k] = n*K + k + 1
    print(f"K={K}: for A one-hot at k=i (all rows), y[m][n] should be w[n][i] = n*K+i+1")
    for i in range(K):
        x = torch.zeros(M, K, device=dev); x[:, i] = 1.0
        y = ops.linear(x, w).cpu().numpy()
        # decode: y[m][n] - (n*K + 1) = j (the k index of the B element paired with A's k=i), if a single product
        j = y - (np.arange(N)[None, :] * K + 1)
        rows = [0, 1, 15, 16, 31]
        print(f"  i={i:2d}: " + "  ".join(f"m{r}:" + ",".join(f"{j[r, c]:.0f}" for c in (0, 1, 15, 16, 31)) for r in rows))
