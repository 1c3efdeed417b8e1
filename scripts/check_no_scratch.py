"""Build-time audit (csrc/Makefile): no kernel of the given object files may use scratch (private_segment_fixed_size > 0 = the register
allocator spilled).  A fused-epilogue experiment of round 3 silently cost the bf16 GEMM kernels 84 bytes per lane this way.
Usage: python3 scripts/check_no_scratch.py gemm.o gemm_p1.o ...   (unbundles the gfx950 code object with llvm-objdump --offloading)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
bad = []
total = 0
for obj in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as tmp:
        src = Path(obj).resolve()
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(src)], cwd=tmp, check=True, capture_output=True)
        # the bundles are written next to the INPUT file, named <input>.<n>.<target>
        cos = list(src.parent.glob(src.name + ".*.hipv4-amdgcn-amd-amdhsa--gfx950"))
        host = list(src.parent.glob(src.name + ".*.host-*"))
        try:
            if not cos:
                sys.exit(f"check_no_scratch: no gfx950 code object in {obj}")
            for co in cos:
                notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
                name = None
                for line in notes.splitlines():
                    m = re.search(r"\.name:\s+(\S+)", line)
                    if m:
                        name = m.group(1)
                    m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
                    if m:
                        total += 1
                        if int(m.group(1)) > 0:
                            bad.append((obj, name, int(m.group(1))))
        finally:
            for f in cos + host:
                f.unlink(missing_ok=True)
if bad:
    for obj, name, size in bad:
        print(f"SCRATCH: {obj}: {name}: {size} bytes per lane", file=sys.stderr)
    sys.exit(1)
print(f"OK: {total} kernels in {len(sys.argv) - 1} object(s), none uses scratch")
