import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib
def run(D, N, dtype, K=5, W=2, L=10, h=0.1):
    rs = np.random.RandomState(0)
    A = rs.standard_normal((D, D)); Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    pot = P.GaussianDense(None, precision=0.5*(Pm+Pm.T), const=0.0, dtype=dtype)
    tdt = torch.float64 if dtype == "float64" else torch.float32
    code = _lib.F64 if dtype == "float64" else _lib.F32
    q = torch.empty((D, N), dtype=tdt, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, D, N, N, 1.0, None, code, 0, q.data_ptr(), None)
    S = max(K, W)
    s = torch.empty((S, D, N), dtype=tdt, device="cuda"); m = torch.empty((S, D, N), dtype=tdt, device="cuda")
    def go(n, it0):
        _lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, s.data_ptr(), m.data_ptr(), None, None, N, N, h, L, n, 1, 7, it0, 0, 1.0, None)
    go(W, 0); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(K, W); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / K
    fl = (L + 1) * 2.0 * D * D * N
    print(json.dumps({"D": D, "N": N, "dtype": dtype, "ms_per_iter": t * 1e3, "TFLOPs": fl / t / 1e12, "rate": L * N / t}), flush=True)
if __name__ == "__main__":
    cfgs = sys.argv[1:] or ["256:65536:float64", "512:32768:float64", "1024:16384:float64",
                            "1024:16384:float32", "256:65536:float32"]
    for c in cfgs:
        d, n, t = c.split(":")
        run(int(d), int(n), t)
