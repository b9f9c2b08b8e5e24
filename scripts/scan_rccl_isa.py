"""Round-3 verdict, next-round item 3(a): does RCCL's own gfx950 code contain the packed-f32 forms that were measured to misread a
source element when a wave shares a SIMD with a wave of another stream's kernel (profiles/r03_simd_coresidency.md:
`v_pk_{add,mul,fma}_f32` whose LOW half selects the HIGH element of its second / third source, op_sel:[0,1(,0)] / [0,0,1])?  RCCL's
reduce / reduce-scatter kernels run on the communication stream BESIDE this library's GEMM column blocks, attention and row-wise
kernels under tensor parallelism -- they would be the victims.
Extracts the gfx950 code object from the library's compressed fat binary (scripts/extract_fatbin_gfx950.py), disassembles it
(llvm-objdump, ~1.5 min for 107 MB of .text) and applies the rule of tests/test_isa_cpu.py.
Usage: python scripts/scan_rccl_isa.py [librccl.so ...]   (default: the ROCm one this library links + torch's bundled one)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import extract_fatbin_gfx950 as ex

LLVM = "/opt/rocm/lib/llvm/bin"


def scan(lib):
    tmp = tempfile.mkdtemp(prefix="rccl_scan_")
    try:
        objs = ex.main(lib, tmp)
        res = {"library": os.path.realpath(lib), "bytes": os.path.getsize(os.path.realpath(lib)), "gfx950_objects": len(objs), "packed_f32": 0,
               "with_op_sel": 0, "affected": [], "by_op": {}, "reduce_functions_with_packed_f32": 0}
        funcs = set()
        for o in objs:
            p = subprocess.Popen([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", o], stdout=subprocess.PIPE, text=True, errors="replace")
            cur = None
            for line in p.stdout:
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    continue
                t = line.strip().split("//")[0].strip()
                m = re.match(r"v_pk_(add|mul|fma)_f32 ", t)
                if not m:
                    continue
                res["packed_f32"] += 1
                res["by_op"][m.group(1)] = res["by_op"].get(m.group(1), 0) + 1
                funcs.add(cur)
                sel = re.search(r"op_sel:\[([01,]+)\]", t)
                if not sel:
                    continue
                res["with_op_sel"] += 1
                bits = [int(b) for b in sel.group(1).split(",")]
                srcs = re.findall(r"(?:v\[\d+:\d+\]|s\[\d+:\d+\])", t)[1:]
                if any(bits[1:]) and len(set(srcs)) > 1:
                    res["affected"].append((cur, t))
            p.wait()
        res["reduce_functions_with_packed_f32"] = len(funcs)
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    libs = sys.argv[1:]
    if not libs:
        libs = ["/opt/rocm/lib/librccl.so"]
        try:
            import torch
            t = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(t):
                libs.append(t)
        except Exception:
            pass
    for lib in libs:
        r = scan(lib)
        print(f"{r['library']} ({r['bytes']} bytes): {r['gfx950_objects']} gfx950 code object(s), {r['packed_f32']} v_pk_{{add,mul,fma}}_f32 "
              f"({r['by_op']}) in {r['reduce_functions_with_packed_f32']} functions, {r['with_op_sel']} with an op_sel modifier, "
              f"{len(r['affected'])} of the affected form", flush=True)
        for f, t in r["affected"][:20]:
            print("   AFFECTED:", f, "|", t)
