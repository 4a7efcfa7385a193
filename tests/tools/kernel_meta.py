"""Kernel metadata (registers, LDS) of the gfx950 code object inside the built library: the clang offload bundle in the
.hip_fatbin section, the AMDGPU metadata note (msgpack) of the ELF it carries."""
import struct

import msgpack


def kernels(lib_path, arch="gfx950"):
    b = open(lib_path, "rb").read()
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if i < 0:
        raise RuntimeError("no uncompressed offload bundle in " + lib_path)
    n = struct.unpack_from("<Q", b, i + 24)[0]
    off = i + 32
    elf = None
    for _ in range(n):
        o, sz, ts = struct.unpack_from("<QQQ", b, off)
        off += 24
        triple = b[off:off + ts].decode()
        off += ts
        if arch in triple:
            elf = b[i + o:i + o + sz]
    if elf is None:
        raise RuntimeError("no %s code object in %s" % (arch, lib_path))
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    out = {}
    for s in range(shnum):
        _, typ, _, _, offset, size = struct.unpack_from("<IIQQQQ", elf, shoff + s * shentsize)
        if typ != 7:          # SHT_NOTE
            continue
        p = offset
        while p < offset + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12 + ((namesz + 3) & ~3)
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32:   # NT_AMDGPU_METADATA
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = dict(vgpr=k[".vgpr_count"], agpr=k.get(".agpr_count", 0), sgpr=k[".sgpr_count"],
                                           lds=k[".group_segment_fixed_size"], scratch=k[".private_segment_fixed_size"])
    return out


if __name__ == "__main__":
    import sys
    for name, m in sorted(kernels(sys.argv[1]).items()):
        if len(sys.argv) < 3 or sys.argv[2] in name:
            print(name, m)
