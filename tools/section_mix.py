#!/usr/bin/env python3
"""tools/section_mix.py FILE.s [KERNEL-SUBSTRING] -- static instruction mix per marked section of a kernel.
Build the ISA with -DHMX_MARKS -S --cuda-device-only: HMX_MARK(n, id) leaves "; HMXMARK n id" comments; every instruction is
attributed to the last mark above it in layout order (an approximation where the compiler moved code across a mark)."""
import collections
import re
import sys

path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_intra_packedILb1ELi64ELb0ELb0")
cur, sec = None, "-"
tab = collections.defaultdict(collections.Counter)
order = []
for ln in open(path):
    s = ln.strip()
    m = re.match(r"^(_Z\w+):", s)
    if m:
        cur = m.group(1) if want in m.group(1) else None
        sec = "entry"
        continue
    if cur is None or not s:
        continue
    if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
        cur = None
        continue
    m = re.search(r"; HMXMARK (0x[0-9a-f]+|\d+) (\d+)", s)
    if m:
        sec = f"N={int(m.group(1), 0)} s{m.group(2)}"
        if sec not in order:
            order.append(sec)
        continue
    if s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
            "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    tab[sec][kind] += 1
    if sec not in order:
        order.append(sec)
tot = collections.Counter()
for sec in order:
    c = tab[sec]
    tot.update(c)
    print(f"{sec:14s} valu {c['valu']:6d} salu {c['salu']:6d} lds {c['lds']:5d} vmem {c['vmem']:4d} mfma {c['mfma']:4d}")
print(f"{'total':14s} valu {tot['valu']:6d} salu {tot['salu']:6d} lds {tot['lds']:5d} vmem {tot['vmem']:4d} mfma {tot['mfma']:4d}")
