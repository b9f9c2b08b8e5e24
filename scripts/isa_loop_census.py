"""Census of a kernel's disassembly between two code offsets (CPU only): instructions per class.
usage: isa_loop_census.py <disassembly.s> <kernel-substring> [start_off end_off]   (offsets relative to the kernel start, hex)"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")): return op.split()[0]
    if op.startswith("s_"): return "salu"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "valu_trans"
    if op.startswith("v_accvgpr"): return "accvgpr_mov"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    path, sub = sys.argv[1], sys.argv[2]
    lo = int(sys.argv[3], 16) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4], 16) if len(sys.argv) > 4 else 1 << 60
    lines = open(path).read().split("\n")
    start = None
    base = 0
    cls, ops = Counter(), Counter()
    for ln in lines:
        m = re.match(r"^([0-9a-f]+) <(.*)>:", ln)
        if m:
            if start is not None:
                break
            if sub in m.group(2):
                start = True
                base = int(m.group(1), 16)
            continue
        if start is None:
            continue
        m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+):", ln)
        if not m:
            continue
        off = int(m.group(2), 16) - base
        if off < lo or off >= hi:
            continue
        ins = m.group(1)
        cls[classify(ins)] += 1
        ops[ins.split()[0]] += 1
    print(dict(cls))
    print("total", sum(cls.values()), "vector-ALU", cls["valu"] + cls["valu_trans"] + cls["accvgpr_mov"])
    for k, v in ops.most_common(40):
        print(f"  {v:4d} {k}")


if __name__ == "__main__":
    main()
