#!/usr/bin/env python3
"""ISA statistics of the hot instantiations of k_gram_bitslice in a built gkm_gram_bitslice.o (runs where llvm-objdump is):
VGPRs, SGPRs, scratch, static LDS, instruction counts by class.  Used to show that a refactoring of the kernel
source left the generated code alone, and to compare experimental builds.

    python3 tools/isa_stats.py [--object gkmqc_amd/csrc/build/gkm_gram_bitslice.o] [--kernel W,L,D,PK ...] [--dump DIR]
"""
import argparse
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
HOT = ["10,11,3,4", "10,10,3,4", "10,12,4,1"]   # config 2, gkmQC's own shape (600 bp), config 5 (ragged)


def unbundle(obj):
    tmp = tempfile.mkdtemp()
    fat, elf = os.path.join(tmp, "fatbin"), os.path.join(tmp, "dev.elf")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf], stderr=subprocess.DEVNULL)
    return elf


def symbol_for(elf, kernel):
    w, L, d, pk = kernel
    want = "k_gram_bitsliceILi%dELi%dELi%dELi%dE" % (w, L, d, pk)
    out = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-sW", elf]).decode()
    for ln in out.splitlines():
        f = ln.split()
        if f and want in f[-1] and not f[-1].endswith(".kd") and "FUNC" in f:
            return f[-1]
    raise SystemExit("no symbol for %s" % (kernel,))


def stats(elf, sym):
    dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + sym, elf]).decode()
    ops = [ln.split("//")[0].split()[0] for ln in dis.splitlines()
           if re.match(r"^\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)", ln)]
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", elf]).decode()
    meta = {}
    blocks = notes.split("- .agpr_count")
    for b in blocks:
        if re.search(r"\.name:\s+" + re.escape(sym) + r"\s", b):
            for key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                m = re.search(r"\." + key + r":\s+(\d+)", b)
                if m:
                    meta[key] = int(m.group(1))
    c = lambda pred: sum(1 for o in ops if pred(o))   # noqa: E731
    return dict(meta, instructions=len(ops), valu=c(lambda o: o.startswith("v_")), salu=c(lambda o: o.startswith("s_")),
                lds=c(lambda o: o.startswith("ds_")), vmem=c(lambda o: o.startswith(("global_", "buffer_", "flat_"))),
                bitop3=c(lambda o: o.startswith("v_bitop3")), scratch=c(lambda o: o.startswith("scratch_")),
                mfma=c(lambda o: "mfma" in o)), dis


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--object", default=os.path.join(ROOT, "gkmqc_amd", "csrc", "build", "gkm_gram_bitslice.o"))
    ap.add_argument("--kernel", action="append", help="W,L,D,PK (default: the three hot instantiations)")
    ap.add_argument("--dump", help="directory for the disassembly of each kernel")
    args = ap.parse_args()
    elf = unbundle(args.object)
    for k in args.kernel or HOT:
        kern = tuple(int(x) for x in k.split(","))
        sym = symbol_for(elf, kern)
        st, dis = stats(elf, sym)
        print("k_gram_bitslice<%s>: %s" % (k, " ".join("%s=%s" % kv for kv in st.items())))
        if args.dump:
            os.makedirs(args.dump, exist_ok=True)
            open(os.path.join(args.dump, "k_%s.s" % k.replace(",", "_")), "w").write(dis)


if __name__ == "__main__":
    main()
