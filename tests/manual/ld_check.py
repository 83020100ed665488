"""Extended-precision (x87 80-bit) replay of doubling/interaction on the numpy twin, to tell which of
{oracle LU float64, GPU series, GPU strip chains, GPU Gauss-Jordan} is closest to the exact result."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rtamd, helpers
from oracle import momref as mr
LD = np.longdouble

def inv_ld(A):
    A = A.astype(LD).copy(); S, N, _ = A.shape
    X = np.broadcast_to(np.eye(N, dtype=LD), A.shape).copy()
    for k in range(N):
        piv = np.abs(A[:, k:, k]).argmax(axis=1) + k
        for s in range(S):
            if piv[s] != k:
                A[s, [k, piv[s]]] = A[s, [piv[s], k]]; X[s, [k, piv[s]]] = X[s, [piv[s], k]]
        d = A[:, k, k][:, None].copy()
        A[:, k] /= d; X[:, k] /= d
        f = A[:, :, k].copy(); f[:, k] = 0
        A -= f[:, :, None] * A[:, k][:, None, :]
        X -= f[:, :, None] * X[:, k][:, None, :]
    return X

def run(model, ld):
    sc = helpers.oracle_scene(model)
    if ld:
        z3 = lambda N, S: np.zeros((S, N, N), dtype=LD); z2 = lambda N, S: np.zeros((S, N), dtype=LD)
        mr.make_added_layer = lambda N, S: mr.AddedLayer(z3(N, S), z3(N, S), z3(N, S), z3(N, S), z2(N, S), z2(N, S))
        mr.make_composite_layer = lambda N, S: mr.CompositeLayer(z3(N, S), z3(N, S), z3(N, S), z3(N, S), z2(N, S), z2(N, S))
        mr.batch_inv = inv_ld
    return mr.rt_run(sc)

gpu = np.load(sys.argv[1])
for tot in (2.0, 8.0):
    m = rtamd.scenes.make_scene(3, 33, 4, 8, seed=77, aerosol_total=tot, aerosol_p0=600.0, aerosol_σp=200.0, absorption=False)
    import importlib; importlib.reload(mr)
    R64, T64 = run(m, False)
    Rx, Tx = run(m, True)
    print("tot", tot, "dtype of ext run:", Rx.dtype)
    def rel(X, Xr): return float((np.abs(X - Xr) / np.abs(Xr[:, :1, :])).max())
    print("  oracle f64 (LU)   vs ext:  R %.2e  T %.2e" % (rel(R64, Rx), rel(T64, Tx)))
    for inv, nm in ((0, "GPU strip chains "), (2, "GPU series (LDS) "), (1, "GPU Gauss-Jordan ")):
        print("  %s vs ext:  R %.2e  T %.2e   | vs oracle: R %.2e T %.2e" % (nm, rel(gpu[f"R_{tot}_{inv}"], Rx), rel(gpu[f"T_{tot}_{inv}"], Tx),
              rel(gpu[f"R_{tot}_{inv}"], R64), rel(gpu[f"T_{tot}_{inv}"], T64)))
