"""Instruction mix per basic block of one kernel in a device assembly file (hipcc --cuda-device-only -S):
python tools/asm_blocks.py file.s '<demangled kernel name substring>'
prints, per label-delimited block with at least one MFMA or 30 instructions: #valu (non-MFMA v_*), #mfma, #salu, #vmem, #lds, and
whether a backward branch targets it (loop head)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = None
for i, l in enumerate(txt):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        dn = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if want in dn:
            start = i
            break
assert start is not None, "kernel not found"
blocks, cur, name = [], [], "entry"
for l in txt[start + 1:]:
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
        continue
    t = l.strip()
    if t and not t.startswith((";", ".", "//")):
        cur.append(t)
blocks.append((name, cur))
order = {n: k for k, (n, _) in enumerate(blocks)}
heads = set()
for k, (n, ins) in enumerate(blocks):
    for t in ins:
        m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in order and order[tgt] <= k:
                heads.add(tgt)
tot = dict(valu=0, mfma=0, salu=0, vmem=0, lds=0)
for n, ins in blocks:
    c = dict(valu=0, mfma=0, salu=0, vmem=0, lds=0, trans=0)
    for t in ins:
        op = t.split()[0]
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if re.match(r"v_(exp|rcp|rsq|sqrt|log|sin|cos)", op): c["trans"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): c["vmem"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
    for k in tot: tot[k] += c[k]
    if c["mfma"] or len(ins) >= 30:
        print(f"{n:>12} {'LOOP' if n in heads else '    '} n={len(ins):5d} valu={c['valu']:4d} (trans {c['trans']:3d}) mfma={c['mfma']:4d} salu={c['salu']:4d} vmem={c['vmem']:3d} lds={c['lds']:3d}")
print("total", tot)
