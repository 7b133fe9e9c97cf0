"""Hot basic blocks of one kernel in a hipcc -save-temps .s file: per block the counts by instruction class and the op histogram.
   python tools/probe/isa_blocks.py file.s kernel_substring [min_instructions]"""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 150
txt = open(path).read()
for m in re.finditer(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(pat), txt, re.S | re.M):
    print(m.group(1)[:100])
    blocks, cur, name = [], [], "entry"
    for l in m.group(2).split("\n"):
        if re.match(r"^\.LBB\S+:", l):
            blocks.append((name, cur)); name = l.split(":")[0]; cur = []
        else:
            s = l.strip()
            if s and not s.startswith((".", ";")):
                cur.append(s)
    blocks.append((name, cur))
    for n, b in blocks:
        nl = sum(1 for x in b if x.startswith(("buffer_load", "global_load")))
        nm = sum(1 for x in b if x.startswith("v_mfma"))
        if nl >= 8 or nm >= 8 or len(b) >= minlen:
            c = collections.Counter()
            for x in b:
                op = x.split()[0]
                k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
                     "vmem" if op.startswith(("buffer", "global", "scratch")) else "wait" if op.startswith("s_waitcnt") else
                     "nop" if op.startswith("s_nop") else "salu")
                c[k] += 1
            print(" ", n, len(b), dict(c))
            print("     ", collections.Counter(x.split()[0] for x in b if x.startswith("v_") and not x.startswith("v_mfma")).most_common(14))
