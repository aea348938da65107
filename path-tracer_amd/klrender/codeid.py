"""Identity of the device code inside a built libklrender.so: one hash per kernel, taken over the kernel's machine code
as it lies in the library file. Measurements that are kept in files (per-ray instruction and traffic figures from
counter passes) carry the hashes of the kernels they were taken on; a reader compares them with the library it loaded
and knows whether the figures still describe the code that runs.

Pure Python over the file's bytes (ELF64 section table -> .hip_fatbin -> clang offload bundles -> the gfx950 code
object's symbol table); nothing is executed and no GPU is needed."""
import hashlib
import struct

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(b, base=0):
    """[(name, type, addr, offset, size, link, entsize)] of the ELF64 image that starts at b[base]"""
    if b[base:base + 4] != b"\x7fELF" or b[base + 4] != 2:
        raise ValueError("not an ELF64 image")
    shoff, = struct.unpack_from("<Q", b, base + 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, base + 0x3A)
    raw = []
    for i in range(shnum):
        name, typ, _flags, addr, off, size, link, _info, _align, entsize = struct.unpack_from("<IIQQQQIIQQ", b, base + shoff + i * shentsize)
        raw.append((name, typ, addr, off, size, link, entsize))
    stroff = raw[shstrndx][3]

    def cstr(o):
        e = b.index(b"\0", base + stroff + o)
        return b[base + stroff + o:e].decode()
    return [(cstr(r[0]),) + r[1:] for r in raw]


def _code_objects(b):
    """(triple, offset, size) of every device code object bundled in the library image"""
    out = []
    for name, _t, _a, off, size, _l, _e in _sections(b):
        if name != ".hip_fatbin":
            continue
        pos = off
        while True:
            pos = b.find(BUNDLE_MAGIC, pos, off + size)
            if pos < 0:
                break
            n, = struct.unpack_from("<Q", b, pos + len(BUNDLE_MAGIC))
            p = pos + len(BUNDLE_MAGIC) + 8
            for _ in range(n):
                eoff, esize, tlen = struct.unpack_from("<QQQ", b, p)
                triple = b[p + 24:p + 24 + tlen].decode()
                p += 24 + tlen
                if esize and "amdgcn" in triple:
                    out.append((triple, pos + eoff, esize))
            pos += len(BUNDLE_MAGIC)
    return out


def kernel_hashes(path):
    """{mangled kernel name: first 16 hex digits of the SHA-1 of its machine code} for every kernel in the library at `path`"""
    b = open(path, "rb").read()
    out = {}
    for _triple, base, _size in _code_objects(b):
        secs = _sections(b, base)
        for name, typ, _addr, off, size, link, entsize in secs:
            if typ != 2 or not entsize:      # SHT_SYMTAB
                continue
            stroff = secs[link][3]
            for i in range(size // entsize):
                st_name, st_info, _other, shndx, value, st_size = struct.unpack_from("<IBBHQQ", b, base + off + i * entsize)
                if (st_info & 0xF) != 2 or not st_size or shndx == 0 or shndx >= len(secs):      # STT_FUNC with a body
                    continue
                e = b.index(b"\0", base + stroff + st_name)
                sym = b[base + stroff + st_name:e].decode()
                sec = secs[shndx]
                start = base + sec[3] + (value - sec[2])
                out[sym] = hashlib.sha1(b[start:start + st_size]).hexdigest()[:16]
    return out


def pick(hashes, prefix):
    """the hash of the one kernel whose mangled name starts with `prefix` (None if there is none or several)"""
    hit = [h for k, h in hashes.items() if k.startswith(prefix)]
    return hit[0] if len(hit) == 1 else None


def traversal_types(bvh_stats):
    """which primitive types a scene's tree holds, as the traversal launcher picks its kernel instance: bit 0 spheres,
    bit 1 quads, bit 2 triangles (flat primitives are kept out of the tree)"""
    return ((1 if bvh_stats.spheres > bvh_stats.flat_spheres else 0) | (2 if bvh_stats.quads > bvh_stats.flat_quads else 0) |
            (4 if bvh_stats.tris > bvh_stats.flat_tris else 0))


def render_kernels(types, mode=1):
    """mangled-name prefixes of the four kernels of a render step whose tree holds `types` (path-traced mode by default)"""
    return {"intersect": "_ZN2kl15k_intersect_bvhILi0ELj%dEE" % (types or 7), "shade": "_ZN2kl7k_shadeILi%dEE" % mode,
            "generate": "_ZN2kl10k_generate", "accumulate": "_ZN2kl12k_accumulate"}


def render_code_identity(path, types, mode=1):
    """{"intersect" | "shade" | "generate" | "accumulate": hash of that kernel in the library at `path`}"""
    h = kernel_hashes(path)
    return {k: pick(h, p) for k, p in render_kernels(types, mode).items()}
