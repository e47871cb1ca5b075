#!/usr/bin/env python3
"""Issue-cost model of the hot kernel from its ISA (runs where llvm-objdump is: the build container).

On gfx950 a SIMD issues the plain integer VALU operations (v_and / v_or / v_xor / v_add / v_sub / v_not / v_mov /
v_lshrrev / v_ashrrev / v_bitop3 on VGPRs, literals and inline constants) about every 2.2 cycles and EVERYTHING ELSE
-- v_lshlrev, v_bfe, v_mad_u32_u24, v_mul_*, v_min/max, v_min3, v_sad, v_alignbit, v_ffbl, v_bcnt, v_mbcnt, compares,
v_cndmask, SDWA / DPP forms, packed 16-bit operations, and any instruction with an SGPR source -- about every 4.2
(tools/valu_ops.hip, profiles/r3_valu_ops.txt).  This script disassembles one k_gram_bitslice instantiation, prices every
VALU instruction with those two rates, and splits the kernel into the counting loop (everything outside the trips,
per shift) and one trip (between the ring read that precedes the first v_ffbl_b32 and the ds_xor_b32 that ends it).

    python3 tools/issue_model.py [--object gkmqc_amd/csrc/build/gkm_device.o] [--kernel 10,11,3,0] [--shifts-per-block 4]
        [--waves N --T len --trips N --ms measured]

With --waves/--T/--trips it predicts the kernel time as (shifts x counting cost + trips x trip cost) / (1024 SIMDs x
clock) and prints it beside --ms.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FULL = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_not_b32", "v_mov_b32", "v_bitop3_b32", "v_add_f32", "v_fma_f32"}
C_FULL, C_HALF = 2.2, 4.2      # cycles per wave64 instruction and SIMD (profiles/r3_valu_ops.txt)


def disassemble(obj, kernel):
    tmp = tempfile.mkdtemp()
    elf = os.path.join(tmp, "dev.elf")
    fat = os.path.join(tmp, "fatbin")   # the host object carries the device code objects as a bundle in .hip_fatbin
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf], stderr=subprocess.DEVNULL)
    w, L, d, pk = kernel
    sym = "_Z15k_gram_bitsliceILi%dELi%dELi%dELi%dELi0EEv6BsArgs" % (w, L, d, pk)
    out = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + sym, elf]).decode()
    lines = []
    for ln in out.splitlines():
        ln = ln.split("//")[0].strip()
        if re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", ln):
            lines.append(ln)
    return sym, lines


def classify(ln):
    """-> None (not VALU), 'F' or 'H'."""
    op = ln.split()[0]
    if not op.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith(("_sdwa", "_dpp")):
        return "H"
    if base not in FULL:
        return "H"
    operands = ln[len(op):].split(",")[1:]          # sources only
    for o in operands:
        o = o.strip().split()[0] if o.strip() else ""
        if re.match(r"^(s\d+|s\[|vcc|exec|m0|ttmp)", o):
            return "H"                               # an SGPR source halves the rate of any instruction
    return "F"


def cost(lines):
    f = sum(1 for x in lines if classify(x) == "F")
    h = sum(1 for x in lines if classify(x) == "H")
    return f, h, f * C_FULL + h * C_HALF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--object", default=os.path.join(ROOT, "gkmqc_amd", "csrc", "build", "gkm_device.o"))
    ap.add_argument("--kernel", default="10,11,3,0", help="W,L,D,PK of the instantiation")
    ap.add_argument("--shifts-per-block", type=int, default=4, help="GKM_BS_DU")
    ap.add_argument("--waves", type=float, default=None)
    ap.add_argument("--T", type=float, default=None, help="(mean) column length: a wave sweeps 2 T shifts")
    ap.add_argument("--trips", type=float, default=None, help="trips per launch (data-dependent: hits / 64 + re-pushes)")
    ap.add_argument("--ms", type=float, default=None, help="measured kernel time to print beside the prediction")
    ap.add_argument("--clock", type=float, default=2.37e9)
    a = ap.parse_args()
    sym, lines = disassemble(a.object, [int(x) for x in a.kernel.split(",")])
    # trips: every copy runs from the ring read (3 x ds_read2st64_b32) before a v_ffbl_b32 to the ds_xor_b32 after it
    ffbl = [i for i, x in enumerate(lines) if x.startswith("v_ffbl_b32")]
    starts = []
    for i in ffbl:
        if starts and i - starts[-1][1] < 40:
            starts[-1][1] = i
            continue
        j = i
        while j > 0 and not lines[j].startswith("ds_read2st64_b32"):
            j -= 1
        while j > 0 and lines[j - 1].startswith(("ds_read2st64_b32", "v_and_b32", "v_add_lshl_u32")):
            j -= 1
        starts.append([j, i])
    copies = []
    for j, i in starts:
        k = i
        while k < len(lines) and not lines[k].startswith("ds_xor_b32"):
            k += 1
        copies.append((j, k))
    in_trip = set()
    for j, k in copies:
        in_trip.update(range(j, k + 1))
    trip = lines[copies[0][0]:copies[0][1] + 1] if copies else []
    rest = [x for i, x in enumerate(lines) if i not in in_trip]
    tf, th, tc = cost(trip)
    rf, rh, rc = cost(rest)
    nshift = 2 * a.shifts_per_block      # the loop over shift blocks is instantiated once per strand
    print("%s" % sym)
    print("  %d instructions, %d trip copies" % (len(lines), len(copies)))
    print("  one trip:            %3d full-rate + %3d half-rate VALU = %5.0f cycles (%d LDS, %d vector-memory, %d scalar ALU)"
          % (tf, th, tc, sum(x.startswith("ds_") for x in trip), sum(x.startswith(("global_", "buffer_")) for x in trip),
             sum(x.startswith("s_") for x in trip)))
    print("  outside the trips:   %3d full-rate + %3d half-rate VALU = %5.0f cycles for %d unrolled shifts -> %.1f cycles "
          "per shift (prologue and epilogue included: an upper bound)" % (rf, rh, rc, nshift, rc / nshift))
    if a.waves and a.T and a.trips is not None:
        shifts = a.waves * 2 * a.T
        cyc = shifts * rc / nshift + a.trips * tc
        ms = cyc / 1024 / a.clock * 1e3
        print("  predicted: %.3g shifts x %.1f + %.3g trips x %.0f = %.3g SIMD-cycles -> %.1f ms at %.2f GHz on 1024 SIMDs%s"
              % (shifts, rc / nshift, a.trips, tc, cyc, ms, a.clock / 1e9,
                 "" if a.ms is None else "; measured %.1f ms -> the VALU issue model explains %.0f %% of it" % (a.ms, 100 * ms / a.ms)))


if __name__ == "__main__":
    sys.exit(main())
