"""Per-kernel summary of rocprofv3 passes over ANY command (the generic sibling of pmc_summary.py, which knows bench.py's kernels):
directory layout <root>/kt (kernel trace) and <root>/<pass>/ (one --pmc pass each, separate runs).
    python tools/pmc_by_kernel.py <root> [out.txt] [top_n]"""
import collections, csv, glob, os, re, sys

root = sys.argv[1]
out_txt = sys.argv[2] if len(sys.argv) > 2 else None
top = int(sys.argv[3]) if len(sys.argv) > 3 else 8

def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("hmvit::", "")

dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "kt", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(dur, key=lambda k: -sum(dur[k]))[:top]
total = sum(sum(v) for v in dur.values())
lines = [f"all kernels: {total / 1e3:.2f} ms in the trace"]
for k in names:
    d = dur[k]
    c = {n: sum(v) / len(v) for n, v in acc.get(k, {}).items()}
    lines.append(f"kernel {k}: {len(d)} launches, total {sum(d) / 1e3:.2f} ms ({100 * sum(d) / total:.1f}%), avg {sum(d) / len(d):.1f} us")
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
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        fetch, write = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
        lines.append("  HBM per launch: FETCH_SIZE %.3f GB (x2 wide-read correction = %.3f GB), WRITE_SIZE %.3f GB" % (fetch / 1e9, 2 * fetch / 1e9, write / 1e9))
print("\n".join(lines))
if out_txt:
    open(out_txt, "w").write("\n".join(lines) + "\n")
