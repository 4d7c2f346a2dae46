#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 assembly file (hipcc -S --cuda-device-only): every backward branch closes a
loop; for each loop body that contains a given marker instruction (default v_rsq_f64: the primitive-quartet loops) print the count of
instructions by class.  usage: isa_loops.py file.s <kernel name substring> [marker]"""
import re, sys, collections
src, kern = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "v_rsq_f64"
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and kern in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].strip().startswith(".end_amdhsa_kernel") or re.match(r"^\s*s_endpgm", lines[i]) and False) if False else None
# kernel text ends at the .Lfunc_end label
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
def cls(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64") or op.startswith("v_mul_f64") or op.startswith("v_add_f64"): return "valu_f64"
    if op.startswith("v_pk_") : return "valu_pk"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_rsq") or op.startswith("v_rcp") or op.startswith("v_sqrt"): return "trans"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"): return "valu_sel"
    if op.startswith("v_accvgpr") : return "accvgpr"
    if op.startswith("v_mov") : return "v_mov"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return "other"
for i, l in enumerate(body):
    m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if not m or m.group(2) not in labels or labels[m.group(2)] >= i: continue
    lo = labels[m.group(2)]
    ins = [x.strip().split()[0] for x in body[lo:i + 1] if x.startswith("\t") and not x.strip().startswith(".") and not x.strip().startswith(";")]
    if not any(marker in x for x in ins): continue
    c = collections.Counter(cls(x) for x in ins)
    inner = sum(1 for x in body[lo:i] if re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", x) and labels.get(re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", x).group(2), 1 << 30) < body.index(x) if False)
    print("loop %s..line %d: %d instructions, %d x %s | %s" % (m.group(2), i, len(ins), sum(1 for x in ins if marker in x), marker,
          "  ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1]))))
