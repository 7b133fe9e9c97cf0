"""Ablation probe of the attention kernel at cfg2 shapes (not a test, not the bench): times
hmvit_window_attention with the HMVIT_ATTN_DEBUG switches of csrc/attn.hip."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import _lib
from oracle import hmvit_oracle as O

L, C, H, W, win = 5, 256, 200, 704, 8
P = H * W
dev = "cuda"
torch.manual_seed(0)
x, pw, mode, rl, mask = O.synthetic_scene(L, C, H, W, [1] * L, seed=1)
pw = pw.to(dev)
ainv = torch.empty(L * L, 8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(_lib.lib.hmvit_pair_affines(pw.data_ptr(), ainv.data_ptr(), L * L, H, W, 0.4, 1.0, st), "aff")
q = (torch.randn(L, P, C, device=dev) * 0.5).half()
kv = (torch.randn(L, 1, 2, P, C, device=dev) * 0.5).half()
b_q = torch.zeros(2, C, device=dev); b_kv = torch.zeros(2, 2, 2 * C, device=dev)
bias = torch.randn(C // 32, 7, 64, 4, device=dev)
out = torch.empty(L, P, C, device=dev, dtype=torch.half)
modes = _lib.i32_array([1] * L); cav = _lib.i32_array([1] * L); ego_e = _lib.i32_array([0] * L)

def run(n_ego=L):
    _lib.check(_lib.lib.hmvit_window_attention(q.data_ptr(), kv.data_ptr(), b_q.data_ptr(), b_kv.data_ptr(),
               bias.data_ptr(), ainv.data_ptr(), modes, cav, ego_e, out.data_ptr(), 1, L, n_ego, L, 1, C, H, W, win,
               int(os.environ.get("PART", "0")), 1, 1, st), "attn")

def bench(tag, dbg):
    os.environ["HMVIT_ATTN_DEBUG"] = str(dbg)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:40s} dbg={dbg:#04x}  {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)

VARIANTS = [("default: 8 waves (4 compute + 4 loader)", 0), ("16 waves (8 + 8)", 2), ("1 tap (all visible)", 0x10),
            ("no loads (all masked)", 0x20), ("no compute", 0x40), ("other item order", 0x200), ("no loader priority", 0x800),
            ("L2-resident region", 0x400), ("1 tap + no compute", 0x50), ("no loads + no compute", 0x60),
            ("general loader loop", 0x1000), ("compute priority instead", 0x8800)]
if os.environ.get("ONLY"):
    want = [int(v, 0) for v in os.environ["ONLY"].split(",")]
    VARIANTS = [v for v in VARIANTS if v[1] in want]
for tag, dbg in VARIANTS:
    bench(tag, dbg)

# ---- cycle trace of workgroup 0 (csrc/attn.hip PC_TRACE): slots 0 iter start, 1 taps done,
# 2 loads issued, 3 gather done, 4 after barrier (loader); 5 compute done, 6 before barrier, 7 after
if os.environ.get("TRACE"):
    tr = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
    os.environ["HMVIT_ATTN_DEBUG"] = os.environ.get("TRACE_DBG", "0")
    os.environ["HMVIT_ATTN_TRACE"] = hex(tr.data_ptr())
    run(); torch.cuda.synchronize()
    t = tr.cpu().reshape(64, 8)
    base = int(t[0, 0])
    print("iter  start  taps  issue  gather  barrier | cmp_done cmp_bar_in cmp_bar_out  (cycles since start)")
    for i in range(0, 24):
        r = [int(v) - base if int(v) else -1 for v in t[i]]
        print(f"{i:3d} " + " ".join(f"{v:8d}" for v in r))
    del os.environ["HMVIT_ATTN_TRACE"]

