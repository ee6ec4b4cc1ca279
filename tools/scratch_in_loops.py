#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) instructions sit?  Disassembles minipath_amd/csrc/kernels.gfx950.co, finds the natural
loops of the kernel whose mangled name contains argv[1] (backward branches), and prints every scratch_load / scratch_store with
the number of loops around it and the length of the innermost one.  usage: scratch_in_loops.py KERNEL_SUBSTRING"""
import re, subprocess, sys, os
co = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minipath_amd", "csrc", "kernels.gfx950.co")
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
fn = None; ins = []
for line in txt.splitlines():
    m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
    if m:
        fn = m.group(2) if sys.argv[1] in m.group(2) and fn is None else (fn if ins and False else None)
        if fn is None and ins: break
        continue
    if fn:
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
print(sys.argv[1], len(ins), "instructions")
addr = [a for a, _, _ in ins]
loops = []
for a, op, args in ins:
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"(\d+)$", args.split("//")[0].strip().split()[-1]) if args else None
        try:
            off = int(args.split()[0])
        except Exception:
            continue
        if off >= 32768: off -= 65536
        tgt = a + 4 + off * 4
        if tgt <= a: loops.append((tgt, a))
sc = [(a, op) for a, op, _ in ins if op.startswith("scratch_")]
print(len(loops), "backward branches;", len(sc), "scratch instructions")
for a, op in sc:
    around = [(t, b) for t, b in loops if t <= a <= b]
    inner = min(((b - t) // 4 for t, b in around), default=0)
    print(f"  {a:#x} {op:24s} loops around: {len(around)}  innermost loop length (dwords): {inner}")
