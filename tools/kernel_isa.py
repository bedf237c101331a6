"""The machine code of a kernel as it sits in the in-tree library: bench.py and tools/collect_profiles.py tie a PMC traffic measurement to the
KERNEL it was taken with (sha-256 over the kernel's code bytes + its 64-byte kernel descriptor: registers, LDS, scratch -- without the
descriptor's offset to the code, which moves with the library's layout), not only to the
source tree: a change elsewhere in bifromq_amd/csrc/ (another kernel, a host-side line) leaves the measured kernel's code as it was, and the
measurement stays valid for it; any change of the kernel's own code -- or of where it reaches its constants -- changes the hash.

Pure Python: the gfx950 code object is taken out of the library's .hip_fatbin (an uncompressed clang offload bundle) and its ELF symbol
table is read directly; nothing is executed, no ROCm tool is needed (the GPU boxes have them, the function must not depend on it).

    python tools/kernel_isa.py [libbmq.so]      # prints the hashes of the kernels the bench lines quote
"""
import hashlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bifromq_amd", "libbmq.so")

# the kernels whose traffic the bench lines quote (bench.py: roofline.kernel of the C3 / C2 / C4 legs) -> their (Itanium-mangled) symbols
KERNELS = {
    "k_walk": "_ZN3bmq6k_walkILi512ELi176ELi152ELb0EEEvNS_9BatchArgsE",       # k_walk<512, 176, 152, false>
    "k_expand": "_ZN3bmq8k_expandENS_9BatchArgsE",
    "k_retain_walk": "_ZN3bmq13k_retain_walkILi8ELb0EEEvNS_10RetainArgsENS_9BatchArgsE",  # k_retain_walk<8, false>
}


def device_code_object(lib_path=LIB, arch="gfx950"):
    data = open(lib_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    base = data.find(magic)
    if base < 0:
        raise ValueError("%s: no clang offload bundle" % lib_path)
    (n,) = struct.unpack_from("<Q", data, base + len(magic))
    p = base + len(magic) + 8
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        p += 24
        triple = data[p:p + tl].decode()
        p += tl
        if triple.startswith("hip") and triple.endswith(arch):
            return data[base + off:base + off + size]
    raise ValueError("%s: no %s code object in the bundle" % (lib_path, arch))


def _symbols(elf):
    """name -> (value, size, section index) of an ELF64 little-endian image; section headers as (offset, addr, size) per index"""
    if elf[:4] != b"\x7fELF" or elf[4] != 2 or elf[5] != 1:
        raise ValueError("not an ELF64 LE image")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        _, typ, _, addr, off, size, link, _, _, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append((typ, addr, off, size, link, entsize))
    syms = {}
    for typ, _, off, size, link, entsize in secs:
        if typ != 2:  # SHT_SYMTAB
            continue
        str_off = secs[link][2]
        for k in range(size // entsize):
            name_i, _info, _other, shndx, value, ssize = struct.unpack_from("<IBBHQQ", elf, off + k * entsize)
            end = elf.index(b"\0", str_off + name_i)
            syms[elf[str_off + name_i:end].decode()] = (value, ssize, shndx)
    return syms, secs


def kernel_hashes(lib_path=LIB, kernels=None):
    """{short name: 16 hex digits} for KERNELS (or the given {short name: mangled symbol}); a kernel the library does not hold maps to None"""
    elf = device_code_object(lib_path)
    syms, secs = _symbols(elf)
    out = {}
    for short, mangled in (kernels or KERNELS).items():
        h = hashlib.sha256()
        ok = True
        for name in (mangled, mangled + ".kd"):  # the code, then the kernel descriptor
            if name not in syms:
                ok = False
                break
            value, size, shndx = syms[name]
            _, addr, off, _, _, _ = secs[shndx]
            blob = bytearray(elf[off + (value - addr):off + (value - addr) + size])
            if name.endswith(".kd"):  # kernel_code_entry_byte_offset (bytes 16..23): where the code lies relative to the descriptor -- layout, not code
                blob[16:24] = b"\0" * 8
            h.update(bytes(blob))
        out[short] = h.hexdigest()[:16] if ok else None
    return out


if __name__ == "__main__":
    for k, v in kernel_hashes(sys.argv[1] if len(sys.argv) > 1 else LIB).items():
        print("%-16s %s" % (k, v))
