"""Extracts the gfx950 code object(s) from a shared library whose .hip_fatbin section is a COMPRESSED clang offload bundle (CCOB, zstd) --
librccl.so ships 5 GB of code objects for a dozen targets that way, and this image's llvm-objdump --offloading cannot decompress it.
Streams the zstd frame through libzstd (ctypes; no python zstd module here), reads the bundle's entry table from the first bytes and
keeps only the gfx950 entries.  Usage: extract_fatbin_gfx950.py <lib.so> <outdir>"""
import ctypes as C
import os
import struct
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"


class Buf(C.Structure):
    _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


def stream_decompress(blob: bytes, want):
    """want(offset_in_output, chunk_bytes) -> bool continue"""
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_createDStream.restype = C.c_void_p
    z.ZSTD_decompressStream.argtypes = [C.c_void_p, C.POINTER(Buf), C.POINTER(Buf)]
    z.ZSTD_decompressStream.restype = C.c_size_t
    z.ZSTD_isError.argtypes = [C.c_size_t]
    z.ZSTD_freeDStream.argtypes = [C.c_void_p]
    ds = z.ZSTD_createDStream()
    z.ZSTD_initDStream.argtypes = [C.c_void_p]
    z.ZSTD_initDStream(ds)
    src = C.create_string_buffer(blob, len(blob))
    ib = Buf(C.cast(src, C.c_void_p), len(blob), 0)
    out = C.create_string_buffer(1 << 24)
    pos = 0
    while ib.pos < ib.size:
        ob = Buf(C.cast(out, C.c_void_p), len(out), 0)
        r = z.ZSTD_decompressStream(ds, C.byref(ob), C.byref(ib))
        if z.ZSTD_isError(r):
            raise RuntimeError("zstd error")
        if ob.pos:
            if not want(pos, out.raw[: ob.pos]):
                break
            pos += ob.pos
        if r == 0 and ob.pos == 0:
            break
    z.ZSTD_freeDStream(ds)


def main(lib, outdir):
    os.makedirs(outdir, exist_ok=True)
    fat = os.path.join(outdir, "fatbin.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    d = open(fat, "rb").read()
    os.remove(fat)
    outs = []
    off = 0
    while True:
        off = d.find(b"CCOB", off)
        if off < 0:
            break
        ver, method = struct.unpack_from("<HH", d, off + 4)
        if ver == 3:
            total, usize = struct.unpack_from("<QQ", d, off + 8)
            hdr = 32
        elif ver == 2:
            total, usize = struct.unpack_from("<II", d, off + 8)
            hdr = 24
        else:
            off += 4
            continue
        if method != 1 or total <= hdr or off + total > len(d):
            off += 4
            continue
        blob = d[off + hdr: off + total]
        state = {"head": b"", "entries": None, "files": {}}

        def want(pos, chunk, state=state):
            if state["entries"] is None:
                state["head"] += chunk
                h = state["head"]
                if len(h) < 32:
                    return True
                assert h[:24] == b"__CLANG_OFFLOAD_BUNDLE__", h[:24]
                n = struct.unpack_from("<Q", h, 24)[0]
                p, ents = 32, []
                try:
                    for _ in range(n):
                        eo, es, tl = struct.unpack_from("<QQQ", h, p)
                        triple = h[p + 24: p + 24 + tl].decode()
                        if len(h) < p + 24 + tl:
                            raise struct.error
                        ents.append((eo, es, triple))
                        p += 24 + tl
                except struct.error:
                    return True
                state["entries"] = [e for e in ents if "gfx950" in e[2]]
                state["all"] = [e[2] for e in ents]
                for eo, es, tr in state["entries"]:
                    state["files"][(eo, es, tr)] = bytearray()
                chunk, pos = h, 0
            done = True
            for (eo, es, tr), buf in state["files"].items():
                lo, hi = max(eo, pos), min(eo + es, pos + len(chunk))
                if lo < hi:
                    buf += chunk[lo - pos: hi - pos]
                if len(buf) < es:
                    done = False
            return not done

        stream_decompress(blob, want)
        for i, ((eo, es, tr), buf) in enumerate(state["files"].items()):
            path = os.path.join(outdir, f"bundle{len(outs)}.{tr.replace('/', '_')}.co")
            open(path, "wb").write(bytes(buf))
            outs.append(path)
        print(f"bundle at {off}: {len(state.get('all', []))} entries, targets: {sorted(set(t.split('--')[-1] for t in state.get('all', [])))}", flush=True)
        off += total
    for p in outs:
        print(p, os.path.getsize(p))
    return outs


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
