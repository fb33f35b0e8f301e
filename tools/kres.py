"""Resources of the kernels in a built library (no GPU needed): VGPRs, SGPRs, LDS, scratch, code bytes.
    python tools/kres.py [lib.so | object.o] [name filter ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gunrock_amd", "libgrx.so")
filters = sys.argv[2:]
with tempfile.TemporaryDirectory() as tmp:
    so = os.path.join(tmp, os.path.basename(lib))
    subprocess.run(["cp", lib, so], check=True)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        path = os.path.join(tmp, f)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], check=True, capture_output=True, text=True).stdout
        syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", path], check=True, capture_output=True, text=True).stdout
        size = {}
        for line in syms.splitlines():
            p = line.split()
            if len(p) >= 8 and p[3] == "FUNC":
                size[p[7]] = int(p[2])
        for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?"
                             r"\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)", notes, re.S):
            lds, name, scratch, sg, vg = int(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5))
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            short = re.sub(r"\(.*", "", dem).replace("void ", "").replace("grx::", "")
            if filters and not any(x in short for x in filters):
                continue
            print("%-70s vgpr %3d sgpr %3d lds %6d scratch %4d code %6d" % (short[:70], vg, sg, lds, scratch, size.get(name, 0)))
