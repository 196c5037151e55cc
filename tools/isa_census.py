#!/usr/bin/env python3
"""Instruction census of a kernel's innermost loops from the compiler's assembly:
  hipcc ... -S --cuda-device-only -o k.s raytracer_amd/csrc/rt_kernels.hip
  python tools/isa_census.py k.s _ZN3rta13render_kernelILb1ELb0ELb0ELb1ELb0EEEvNS_12RenderParamsE [min depth]
Per basic block at loop depth >= min depth (default 3: the traversal rounds): instructions by class.
slow = vector instructions that occupy the SIMD for 4.1-4.6 cycles per wave (min/max, compare, select, shift, f64, mad, perm, cvt, lshl_add),
fma32 = f32 add, multiply and multiply-add (2.5 cycles), simple = 32-bit integer add/sub/and/or/xor/mov (2.6 cycles);
measured: profiles/r02_valu_rate.txt, profiles/r02_bank_rate.txt."""
import re
import sys
from collections import Counter

path, kernel = sys.argv[1], sys.argv[2]
min_depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
text = open(path).read()
body = text[text.index("\n" + kernel + ":"):]
body = body[:body.index("s_endpgm")]
SIMPLE = re.compile(r"v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|mov_b32|not_b32)(_e32|_e64|_sdwa|_dpp)?$")
FMA32 = re.compile(r"v_(fma_f32|fmac_f32|mac_f32|mul_f32|add_f32|sub_f32|subrev_f32)(_e32|_e64|_sdwa|_dpp)?$")


def klass(op):
    if op.startswith("v_"):
        return "simple" if SIMPLE.match(op) else "fma32" if FMA32.match(op) else "slow"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


blocks, name, depth = [], "entry", 0
cur = Counter()
fresh = False  # directly after a label: the comment lines that follow may name the loop the block heads
for line in body.splitlines():
    m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", line)
    if m:
        blocks.append([name, depth, cur])
        name, cur = m.group(1), Counter()
        d = re.search(r"(?:Header: |in Loop: Header=\S+ )Depth=(\d+)", m.group(2))
        depth = int(d.group(1)) if d else 0
        fresh = True
        continue
    s = line.strip()
    if fresh and s.startswith(";"):
        d = re.search(r"Loop Header: Depth=(\d+)", s)
        if d:
            depth = max(depth, int(d.group(1)))
        continue
    fresh = False
    if not s or s.startswith(";") or s.startswith("."):
        continue
    cur[klass(s.split()[0])] += 1
blocks.append([name, depth, cur])
cols = ["slow", "fma32", "simple", "lds", "vmem", "salu", "branch", "wait"]
print("%-12s %5s " % ("block", "depth") + " ".join("%7s" % c for c in cols))
tot = Counter()
for name, depth, c in blocks:
    if depth >= min_depth and sum(c.values()):
        print("%-12s %5d " % (name, depth) + " ".join("%7d" % c[k] for k in cols))
        tot.update(c)
print("%-12s %5s " % ("sum", "") + " ".join("%7d" % tot[k] for k in cols))
whole = Counter()
for _, _, c in blocks:
    whole.update(c)
print("%-12s %5s " % ("whole kernel", "") + " ".join("%7d" % whole[k] for k in cols))
