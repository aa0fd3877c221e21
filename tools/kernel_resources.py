"""Registers / LDS / scratch of every gfx950 kernel in libkmx.so (or another code object holder): the code-object notes
of the device binary inside the .hip_fatbin section.  usage: python tools/kernel_resources.py [file] [name filter regex]"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "kmcex_amd", "libkmx.so")
flt = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"k_round|k_slow_fin|k_reorder|k_query<")
data = open(path, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pos = data.find(MAGIC)
if pos < 0:
    sys.exit("no uncompressed offload bundle in " + path + (" (compressed: CCOB)" if b"CCOB" in data else ""))
(n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
off = pos + len(MAGIC) + 8
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off)
    triple = data[off + 24: off + 24 + tl].decode()
    off += 24 + tl
    if "gfx" not in triple:
        continue
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[pos + o: pos + o + sz])
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    print(triple)
    for blk in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm:
            continue
        dem = subprocess.run(["c++filt", nm.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        if not flt.search(dem):
            continue
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
        print("  %-44s vgpr %3s sgpr %3s lds %6s scratch %4s spill %s" % (dem.replace("void ", "")[:44], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count")))
