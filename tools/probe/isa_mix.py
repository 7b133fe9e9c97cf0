"""Instruction mix of one kernel in a hipcc -save-temps .s file: counts by class (static, i.e. per pass through the code, loops
counted once) + registers / scratch.   python tools/probe/isa_mix.py file.s kernel_substring"""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
txt = open(path).read()
# kernels start at "<name>:" after a .globl / .type line and end at s_endpgm ... .Lfunc_end
for m in re.finditer(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(pat), txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    c = collections.Counter()
    ops = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        mm = re.match(r"([a-z_0-9]+)", line)
        if not mm or line.startswith((".", ";")) or line.endswith(":"):
            continue
        op = mm.group(1)
        ops[op] += 1
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem" if not op.startswith("scratch_") else "scratch"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    meta = re.search(r"\.amdhsa_kernel %s.*?\.end_amdhsa_kernel" % re.escape(name), txt, re.S)
    regs = ""
    if meta:
        g = lambda k: (re.search(r"%s (\S+)" % k, meta.group(0)) or [None, "?"])[1]
        regs = f"vgpr {g('.amdhsa_next_free_vgpr')} accum_offset {g('.amdhsa_accum_offset')} sgpr {g('.amdhsa_next_free_sgpr')} scratch {g('.amdhsa_private_segment_fixed_size')} lds {g('.amdhsa_group_segment_fixed_size')}"
    print(name[:90]); print("  ", dict(c)); print("  ", regs)
    print("   top valu:", [(k, v) for k, v in ops.most_common(60) if k.startswith("v_") and not k.startswith("v_mfma")][:24])
