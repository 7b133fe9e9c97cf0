"""Debug aid: persistent attention kernel vs the one-workgroup-per-window kernel on the same f16 inputs."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import _lib
from oracle import hmvit_oracle as O

L, C, H, W, win = 3, 256, 32, 48, 8
P = H * W
dev = "cuda"
torch.manual_seed(0)
x, pw, mode, rl, mask = O.synthetic_scene(L, C, H, W, [1, 0, 1], seed=1, tx_step=5.0, ty_step=-3.0)
pw = pw.to(dev)
ainv = torch.empty(L * L, 8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(_lib.lib.hmvit_pair_affines(pw.data_ptr(), ainv.data_ptr(), L * L, H, W, 0.4, 4.0, st), "aff")
q = (torch.randn(L, P, C, device=dev) * 0.5).half()
kv = (torch.randn(L, 2, 2, P, C, device=dev) * 0.5).half()
b_q = torch.randn(2, C, device=dev) * float(os.environ.get("BQ", "1")); b_kv = torch.randn(2, 2, 2 * C, device=dev) * float(os.environ.get("BKV", "1"))
bias = torch.randn(C // 32, 7, 64, 4, device=dev)
modes = _lib.i32_array([1, 0, 1]); cav = _lib.i32_array([1] * L); ego_e = _lib.i32_array([1, 0, 1])

def run(variant):
    os.environ["HMVIT_ATTN_DEBUG"] = str(variant)
    out = torch.zeros(L, P, C, device=dev, dtype=torch.half)
    _lib.check(_lib.lib.hmvit_window_attention(q.data_ptr(), kv.data_ptr(), b_q.data_ptr(), b_kv.data_ptr(),
               bias.data_ptr(), ainv.data_ptr(), modes, cav, ego_e, out.data_ptr(), 1, L, L, L, 2, C, H, W, win,
               int(os.environ.get("PART", "0")), 1, 1, st), "attn")
    torch.cuda.synchronize()
    return out.float()

a = run(1 | int(os.environ.get("VA", "0")))   # per-window kernel
b = run(int(os.environ.get("VB", "0")))   # persistent kernel
d = (a - b).abs()
print("max |a|", float(a.abs().max()), " max diff", float(d.max()), " mean diff", float(d.mean()))
for e in range(L):
    print(" ego", e, "max diff", float(d[e].max()))
