"""Registers / scratch / LDS of the kernels in the built library whose (demangled) name contains a pattern:
   python tools/dev/kregs.py conv3x3_split [lib]"""
import re, struct, subprocess, sys, tempfile, os
llvm = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1] if len(sys.argv) > 1 else ""
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "../../slr-sfs_amd/lib/libslrsplat.so")
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "stripped")])
data = open(fat, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
start = data.find(magic)
k = 0
while start >= 0:
    n = struct.unpack_from("<Q", data, start + 24)[0]
    p = start + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple and size:
            co = os.path.join(tmp, f"dev{k}.co"); k += 1
            open(co, "wb").write(data[start + off:start + off + size])
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s*(\S+)", blk).group(1)
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                if pat not in dem:
                    continue
                g = lambda key: re.search(rf"\.{key}:\s*(\d+)", blk).group(1)
                agpr = re.match(r"\s*(\d+)", blk).group(1)
                print(f"vgpr {g('vgpr_count'):>4} agpr {agpr:>4} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}  {dem[:120]}")
    start = data.find(magic, start + 1)
