"""Summarise the rocprofv3 passes of tools/probe/profile_all.sh: per-kernel averages of every counter, the
HBM traffic per launch (FETCH_SIZE doubled per the gfx950 wide-read correction of MI355X_MICROARCH.md), and
profiles/pmc_traffic.json for bench.py's roofline.traffic."""
import collections, csv, glob, json, os, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
out_txt = sys.argv[2] if len(sys.argv) > 2 else None
prec = sys.argv[3] if len(sys.argv) > 3 else "f16"
KEYS = ["k_attention_pcs", "k_attention_pc<", "k_attention<float", "k_ln_qkv", "k_out_ffn_qkv", "k_out_ffn_head", "k_gemm<float, false, false, false>",
        "k_gemm<float, false, false, true>", "k_gemm<float, false, true, false>", "k_layernorm"]
PHASE = {"k_attention_pcs": "attention", "k_attention_pc<": "attention", "k_attention<float": "attention", "k_ln_qkv": "ln_qkv",
         "k_out_ffn_qkv": "stage_tail", "k_out_ffn_head": "stage_tail_head", "k_gemm<float, false, false, false>": "qkv_gemm",
         "k_gemm<float, false, false, true>": "out_proj_ffn2", "k_gemm<float, false, true, false>": "ffn1", "k_layernorm": "ln"}

def key_of(name):
    for k in KEYS:
        if k in name:
            return k
    return None

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = key_of(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "kt", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = key_of(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = []
traffic = {}
for k in KEYS:
    if k not in acc:
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    d = dur.get(k, [0])
    lines.append(f"kernel {k}: {len(d)} launches, avg {sum(d) / len(d):.1f} us (min {min(d):.1f}, max {max(d):.1f})")
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        lines.append("  SQ_WAVE_CYCLES %.3e  WAIT_ANY %.0f%%  WAIT_INST_ANY %.0f%%  ACTIVE_INST_ANY %.0f%%  BUSY_CYCLES %.3e" % (
            wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            c.get("SQ_BUSY_CYCLES", 0)))
    lines.append("  INSTS: VALU %.3e  SALU %.3e  LDS %.3e  VMEM_RD %.3e  VMEM_WR %.3e;  MFMA busy cycles %.3e" % tuple(
        c.get(n, 0) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_VALU_MFMA_BUSY_CYCLES")))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        lines.append("  LDS: bank conflict cycles %.3e / active %.3e (%.1f%%), WAIT_INST_LDS %.3e" % (
            c.get("SQ_LDS_BANK_CONFLICT", 0), c["SQ_LDS_IDX_ACTIVE"], 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"],
            c.get("SQ_WAIT_INST_LDS", 0)))
    if c.get("TCC_HIT_sum") is not None:
        h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        tcp, req = c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), c.get("TCP_TCC_READ_REQ_sum", 0)
        lines.append("  L2 hit %.0f%% (hit %.3e miss %.3e); L1: %.3e accesses, %.3e read requests to L2 (hit ~%.0f%%)" % (
            100 * h / max(h + m, 1), h, m, tcp, req, 100 * (1 - req / max(tcp, 1))))
    fetch, write = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
    lines.append("  HBM per launch: FETCH_SIZE %.3f GB (x2 wide-read correction = %.3f GB), WRITE_SIZE %.3f GB" % (fetch / 1e9, 2 * fetch / 1e9, write / 1e9))
    traffic[PHASE[k]] = int(2 * fetch + write)
print("\n".join(lines))
if out_txt:
    open(out_txt, "w").write("\n".join(lines) + "\n")
    # per-precision traffic table for bench.py's roofline.traffic, stamped with the hash of the kernel sources it was measured on
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    path = os.path.join(os.path.dirname(out_txt), "pmc_traffic.json")
    allt = json.load(open(path)) if os.path.exists(path) else {}
    h = bench.kernel_source_hash()
    if allt.get("kernel_source_hash") != h:
        allt = {"kernel_source_hash": h}
    allt[prec] = traffic
    json.dump(allt, open(path, "w"), indent=1)
