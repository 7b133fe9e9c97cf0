import os, sys, time, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hmvit_amd
from hmvit_amd import _lib
M, N, K = 140800, 256, 256
torch.manual_seed(0)
dy = torch.randn(M, N, device="cuda"); a = torch.randn(M, K, device="cuda")
dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
st = torch.cuda.current_stream().cuda_stream
f = _lib.lib.hmvit_gemm_tn
def run():
    _lib.check(f(ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(dw.data_ptr()), ctypes.c_void_p(db.data_ptr()), M, N, K, N, K, ctypes.c_void_p(st)), "gemm_tn")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ref = (dy.double().T @ a.double())
dw.zero_(); run(); torch.cuda.synchronize()
print(os.environ.get("HMVIT_LIB", "shipped"), f"gemm_tn {M}x{N}x{K}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, rel err {float((dw.double() - ref).abs().max() / ref.abs().max()):.2e}")
