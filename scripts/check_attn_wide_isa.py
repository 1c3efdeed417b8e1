"""Audit of the attention and GEMM kernels' ISA (run after every edit of attention*.hip / gemm.hip; CPU only, needs hipcc).  DH = 384:
  * no scratch (a spilled Q fragment reloads every tile),
  * no compiler-generated v_accvgpr_* or MFMA outside the ASMSTART / ASMEND blocks (a0..a191 hold O^T and belong to the asm
    statements of attn_acc_regs.h; the compiler must not allocate accumulator registers of its own),
  * prints the register budget and the instruction mix of the key loop.
Usage: python scripts/check_attn_wide_isa.py [--no-gemm | --attention-only | --gemm4w-isa FILE.s]
(--attention-only: attention.hip alone -- the Makefile runs this after compiling attention.o and fails the build on a violation.)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "algonauts-2025_amd" / "csrc"


def device_isa(sources: list[str]) -> dict[str, str]:
    """gfx950 assembly of each source, the compilations running side by side (device side only)."""
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in sources:
            out = Path(tmp, src.replace(".hip", ".s"))
            procs.append((src, out, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                                                      f"-I{CSRC}", f"-I{ROOT / 'include'}", str(CSRC / src), "-o", str(out)],
                                                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=tmp)))
        isa = {}
        for src, out, proc in procs:
            _, err = proc.communicate()
            if proc.returncode:
                raise SystemExit(f"hipcc failed on {src}:\n{err.decode()[-2000:]}")
            isa[src] = out.read_text()
        return isa


def audit_gemm4w(isa_text: str) -> bool:
    """The one-wave-per-SIMD GEMM kernel (gemm.hip part 3) keeps 64 accumulators in compiler-allocated AGPRs behind "+a" operands of inline-asm
    MFMAs: it must compile without scratch, with all 256 AGPRs, and between the first and the last MFMA of a kernel the compiler must
    neither move accumulators (v_accvgpr_*) nor touch scratch."""
    good = True
    found = re.findall(r"\.amdhsa_kernel (_ZN\S*gemm_nt_4w256\S*)\n(.*?)\.end_amdhsa_kernel", isa_text, re.S)
    if not found:
        print("FAIL: no gemm_nt_4w256 kernel in the ISA")
        return False
    for name, body in found:
        meta = {k: int(v) for k, v in re.findall(r"\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size)\s+(\d+)", body)}
        m = re.search(rf"^{re.escape(name)}:(.*?)^\s*s_endpgm", isa_text, re.S | re.M)
        code = m.group(1).splitlines() if m else []
        # basic blocks (split at labels and branches): a block that issues MFMAs is K-loop code and must hold no accumulator moves or scratch
        blocks, cur_block = [], []
        for ln in code:
            t = ln.strip()
            if t.startswith(".LBB") or re.match(r"s_c?branch", t):
                blocks.append(cur_block)
                cur_block = []
            else:
                cur_block.append(t)
        blocks.append(cur_block)
        n_mfma = sum(1 for ln in code if "v_mfma" in ln)
        bad = [t for blk in blocks if any("v_mfma" in t for t in blk) for t in blk if re.search(r"\bv_accvgpr|scratch_", t)] if n_mfma else ["no MFMA found"]
        line = f"gemm_nt_4w256 [{name[-34:]}]: {meta}, {n_mfma} MFMAs"
        if meta.get("private_segment_fixed_size", 1) or meta.get("next_free_vgpr", 9999) > 512 or meta.get("next_free_vgpr", 0) - meta.get("accum_offset", 0) < 256 or bad:
            print("FAIL (scratch, registers, or accumulator traffic inside the K loop):", line, *bad[:5], sep="\n  ")
            good = False
        else:
            print(line)
    return good


if "--gemm4w-isa" in sys.argv[1:]:   # audit a compiled ISA file (the Makefile passes the -save-temps output of gemm.hip part 3)
    ok4 = audit_gemm4w(Path(sys.argv[sys.argv.index("--gemm4w-isa") + 1]).read_text())
    print("OK: one-wave-per-SIMD GEMM kernels keep their accumulators in place" if ok4 else "gemm4w audit FAILED")
    sys.exit(0 if ok4 else 1)

ATTN_ONLY = "--attention-only" in sys.argv[1:]
WITH_GEMM = "--no-gemm" not in sys.argv[1:] and not ATTN_ONLY   # gemm.hip takes two minutes to compile (its role instantiations); the unit test skips it
ISA = device_isa(["attention.hip"] + ([] if ATTN_ONLY else ["attention_d64.hip"]) + (["gemm.hip"] if WITH_GEMM else []))
text = ISA["attention.hip"]
def audit(kernel: str, max_regs: int) -> bool:
    m = re.search(rf"^(_ZN\S*{kernel}\S*):.*?\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
    if m is None:
        print(f"FAIL: kernel {kernel} not found in the ISA")
        return False
    body = m.group(2)
    meta = {k: int(v) for k, v in re.findall(r"\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size)\s+(\d+)", body)}
    in_asm, bad, mix = False, [], {"v_mfma": 0, "ds_read": 0, "global_load_lds": 0, "s_barrier": 0, "v_exp": 0}
    for ln in body.splitlines():
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        for k in mix:
            if re.search(rf"\b{k}", ln):
                mix[k] += 1
        if not in_asm and re.search(r"\bv_accvgpr|\bv_mfma|scratch_", ln):
            bad.append(ln.strip())
    print(f"{kernel}: registers {meta}; instruction mix (whole kernel) {mix}")
    if meta.get("private_segment_fixed_size", 1) or bad or meta.get("next_free_vgpr", 9999) > max_regs:
        print("FAIL: scratch, too many registers, or compiler traffic on accumulator registers outside the asm blocks:", *bad[:10], sep="\n  ")
        return False
    return True


ok = audit("attn_fwd_wide384", 512) & audit("attn_fwd_ksplit384", 256)

# The DH = 64 kernels (attention_d64.hip) and the two forms of the 256^2 GEMM use compiler-allocated registers throughout; what must hold
# is the budget of two waves per SIMD WITHOUT scratch: a spilled accumulator or Q fragment is reloaded every key tile / K-tile.
def no_scratch(source: str, kernels: list[str]) -> bool:
    isa = ISA[source]
    good = True
    for kernel in kernels:
        found = re.findall(rf"\.amdhsa_kernel (_ZN\S*{kernel}\S*)\n(.*?)\.end_amdhsa_kernel", isa, re.S)
        if not found:
            print(f"FAIL: kernel {kernel} not found in the ISA of {source}")
            good = False
        for name, body in found:
            meta = {k: int(v) for k, v in re.findall(r"\.amdhsa_(next_free_vgpr|private_segment_fixed_size)\s+(\d+)", body)}
            line = f"{kernel} [{name[-40:]}]: {meta}"
            if meta.get("private_segment_fixed_size", 1) or meta.get("next_free_vgpr", 9999) > 256:
                print("FAIL (scratch or more than 256 registers):", line)
                good = False
            else:
                print(line)
    return good


if not ATTN_ONLY:
    ok &= no_scratch("attention_d64.hip", ["attn_fwd_d64_kernel", "attn_fwd_d64_pair_kernel"])
if WITH_GEMM:
    ok &= no_scratch("gemm.hip", ["gemm_nt_256x256x64"])
if not ok:
    sys.exit(1)
print("OK: no scratch, no compiler accumulator-register traffic")
