#!/usr/bin/env python3
"""Issue-cost model of the hot kernel from its ISA (runs where llvm-objdump is: the build container).

On gfx950 a SIMD issues the plain integer VALU operations (v_and / v_or / v_xor / v_add / v_sub / v_not / v_mov /
v_lshrrev / v_ashrrev / v_bitop3 on VGPRs, literals and inline constants) about every 2.2 cycles and EVERYTHING ELSE
-- v_lshlrev, v_bfe, v_mad_u32_u24, v_mul_*, v_min/max, v_min3, v_sad, v_alignbit, v_ffbl, v_bcnt, v_mbcnt, compares,
v_cndmask, SDWA / DPP forms, packed 16-bit operations, and any instruction with an SGPR source -- about every 4.2
(tools/valu_ops.hip, profiles/r3_valu_ops.txt).  This script disassembles one k_gram_bitslice instantiation, prices every
VALU instruction with those two rates, and splits the kernel into the counting loop (everything outside the trips,
per shift) and one trip (between the s_setprio that raises the wave's priority and the s_setprio 0 that ends it).

    python3 tools/issue_model.py [--object gkmqc_amd/csrc/build/gkm_gram_bitslice.o] [--kernel 10,11,3,4] [--shifts-per-block 4]
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
# cycles per wave64 instruction and SIMD (profiles/r3_valu_ops.txt).  C_SGPR: a full-rate opcode with an SGPR source costs
# 4.2 in a stream made of nothing else, but NOTHING extra where VGPR-only instructions sit around it ("mix 8: 2 SGPR-operand,
# adjacent" 2.30-2.33 against 2.30-2.32 without any), which is how the kernel uses them: two per word among twelve others.
# C_HALF: one half-rate instruction among seven full-rate ones costs 4.0-4.1, four alternating with full-rate ones 3.8 each.
# Round 4: with the trips prioritised (s_setprio) the kernel ran FASTER than 2.2 / 4.2 allow (issue cycles / SIMD-cycles
# 1.05-1.12), so the prices are the lower ends of what tools/valu_ops measured: 2.1 for the full-rate opcodes (v_add, v_sub,
# v_not, v_mov, v_lshrrev: 2.07-2.11; v_bitop3 with two VGPR sources 2.27) and 3.8 for a half-rate opcode among full-rate
# ones ("four alternating with full-rate ones 3.8 each").  The model is good to about +-5 %.
C_FULL, C_SGPR, C_HALF = 2.1, 2.1, 3.8
# Round 5 (sensitivity probes, profiles/r5_trip_sensitivity.txt: builds with 16 more VALU instructions / 4 more LDS
# operations per trip): inside a trip one more VALU instruction costs 1.9 (full-rate) to 2.1 ("half-rate") SIMD-cycles --
# the second pass of a half-rate opcode hides among the trip's waits -- and one more LDS instruction costs 5.8: THREE VALU
# instructions.  The VALU-only prices above therefore UNDER-price a trip (12-15 LDS instructions) and over-price its
# half-rate opcodes; `cycles_all` below is the round-5 price: every VALU instruction of a trip 2.0, of the counting loop as
# above, every LDS instruction 5.8, the vector-memory gather ~14 (what one more gather cost the first round-5 hit path).
C_TRIP_VALU, C_LDS, C_VMEM = 2.0, 5.8, 14.0


def disassemble(obj, kernel):
    tmp = tempfile.mkdtemp()
    elf = os.path.join(tmp, "dev.elf")
    fat = os.path.join(tmp, "fatbin")   # the host object carries the device code objects as a bundle in .hip_fatbin
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf], stderr=subprocess.DEVNULL)
    w, L, d, pk = kernel
    sym = "_Z15k_gram_bitsliceILi%dELi%dELi%dELi%dEEv6BsArgs" % (w, L, d, pk)
    out = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + sym, elf]).decode()
    lines, addrs = [], []
    for raw in out.splitlines():
        ln = raw.split("//")[0].strip()
        if re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", ln):
            lines.append(ln)
            m = re.search(r"//\s*([0-9A-Fa-f]+):", raw)
            addrs.append(int(m.group(1), 16) if m else -1)
    disassemble.addrs = addrs
    return sym, lines


def loops(lines, addrs):
    """Backward branches -> [(first index, last index)] of the loop bodies, outermost first."""
    at = {a: i for i, a in enumerate(addrs)}
    out = []
    for i, ln in enumerate(lines):
        if ln.startswith(("s_cbranch", "s_branch")):
            off = int(ln.split()[-1])
            if off >= 32768:
                off -= 65536
            tgt = addrs[i] + 4 + 4 * off
            if tgt <= addrs[i] and tgt in at:
                out.append((at[tgt], i))
    return sorted(out, key=lambda b: b[0] - b[1])


def classify(ln):
    """-> None (not VALU), 'F' (full rate), 'S' (full-rate opcode, SGPR source) or 'H' (half rate)."""
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
            return "S"                               # an SGPR source: up to half the rate
    return "F"


def cost(lines):
    f = sum(1 for x in lines if classify(x) == "F")
    sg = sum(1 for x in lines if classify(x) == "S")
    h = sum(1 for x in lines if classify(x) == "H")
    return f, (sg, h), f * C_FULL + sg * C_SGPR + h * C_HALF


def analyse(obj, kernel, du):
    sym, lines = disassemble(obj, kernel)
    # trips: every copy runs from the s_setprio that raises the wave's priority to the s_setprio 0 that ends it (both trip
    # kinds: the single-hit one with its 3 x ds_read2st64_b32 ... ds_xor_b32, the same-length variant's group trip)
    copies = []
    start = None
    for i, x in enumerate(lines):
        if x.startswith("s_setprio"):
            if not x.split()[1].startswith("0"):
                start = i
            elif start is not None:
                copies.append((start, i))
                start = None
    in_trip = set()
    for j, k in copies:
        in_trip.update(range(j, k + 1))
    trip = lines[copies[0][0]:copies[0][1] + 1] if copies else []
    # the counting loop: the loop over blocks of `du` shifts -- the smallest loop body that holds the scalar loads of the
    # column words and (trips taken out) more than 80 v_bitop3_b32 per shift
    nshift, rest = None, None
    for b0, b1 in reversed(loops(lines, disassemble.addrs)):          # innermost first
        body = [x for i, x in enumerate(lines[b0:b1 + 1], b0) if i not in in_trip]
        if sum(x.startswith("v_bitop3_b32") for x in body) >= 60 * du and any(x.startswith("s_load_dwordx") for x in body):
            nshift, rest = du, body
            break
    if rest is None:      # no loop found: everything outside the trips, an upper bound
        nshift, rest = 2 * du, [x for i, x in enumerate(lines) if i not in in_trip]
    tf, (ts, th), tc = cost(trip)
    rf, (rs, rh), rc = cost(rest)
    t_lds, t_vmem = sum(x.startswith("ds_") for x in trip), sum(x.startswith(("global_", "buffer_")) for x in trip)
    r_lds = sum(x.startswith("ds_") for x in rest) / nshift
    return {"symbol": sym, "instructions": len(lines), "trip_copies": len(copies),
            "trip": {"full_rate": tf, "sgpr_operand": ts, "half_rate": th, "cycles": tc, "lds": t_lds,
                     "vmem": t_vmem, "salu": sum(x.startswith("s_") for x in trip),
                     "cycles_all": (tf + ts + th) * C_TRIP_VALU + t_lds * C_LDS + t_vmem * C_VMEM},
            "per_shift": {"full_rate": rf / nshift, "sgpr_operand": rs / nshift, "half_rate": rh / nshift, "cycles": rc / nshift,
                          # (the counting loop's LDS instructions are the EXEC-masked pushes of a few lanes: their cost was not
                          # probed and is left out -- pricing them like a trip's full-wave operations over-counts)
                          "lds": r_lds, "cycles_all": rc / nshift,
                          "note": "the loop over blocks of %d shifts, trips taken out, / %d" % (nshift, nshift)}}


def from_pmc(model, pmc, mean_T, kernel_ns):
    """Dynamic split from what rocprofv3 counted: waves (= work items) sweep 2 T shifts each; the VALU instructions that
    the shifts do not account for are trips.  -> predicted issue cycles against the cycles the launch had."""
    p = pmc["per_launch"]
    shifts = p["SQ_WAVES"] * 2.0 * mean_T
    per_shift = model["per_shift"]["full_rate"] + model["per_shift"]["sgpr_operand"] + model["per_shift"]["half_rate"]
    per_trip = model["trip"]["full_rate"] + model["trip"]["sgpr_operand"] + model["trip"]["half_rate"]
    trips = max(0.0, (p["SQ_INSTS_VALU"] - shifts * per_shift) / per_trip)
    issue = shifts * model["per_shift"]["cycles"] + trips * model["trip"]["cycles"]
    clock = p["GRBM_GUI_ACTIVE"] / 8.0 / (kernel_ns * 1e-9)           # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    have = kernel_ns * 1e-9 * clock * 1024                              # SIMD-cycles of the launch
    issue_all = shifts * model["per_shift"]["cycles_all"] + trips * model["trip"]["cycles_all"]
    return {"shifts": shifts, "trips": trips, "valu_in_trips": trips * per_trip / p["SQ_INSTS_VALU"],
            "issue_cycles": issue, "simd_cycles": have, "clock_GHz": clock / 1e9, "issue_frac": issue / have,
            "issue_cycles_in_trips": trips * model["trip"]["cycles"] / issue,
            # round 5's prices (LDS instructions included): what share of the launch's SIMD-cycles they account for
            "issue_cycles_all": issue_all, "issue_frac_all": issue_all / have,
            "issue_cycles_all_in_trips": trips * model["trip"]["cycles_all"] / issue_all}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--object", default=os.path.join(ROOT, "gkmqc_amd", "csrc", "build", "gkm_gram_bitslice.o"))
    ap.add_argument("--kernel", default="10,11,3,4", help="W,L,D,PK of the instantiation")
    ap.add_argument("--shifts-per-block", type=int, default=4, help="GKM_BS_DU")
    ap.add_argument("--waves", type=float, default=None)
    ap.add_argument("--T", type=float, default=None, help="(mean) column length: a wave sweeps 2 T shifts")
    ap.add_argument("--trips", type=float, default=None, help="trips per launch (data-dependent: hits / 64 + re-pushes)")
    ap.add_argument("--ms", type=float, default=None, help="measured kernel time to print beside the prediction")
    ap.add_argument("--clock", type=float, default=2.37e9)
    ap.add_argument("--round", default=None, help="e.g. r3: combine with profiles/<round>_pmc_<workload>.json and "
                    "<round>_kernel_stats_<workload>.csv for c2, peaks, c5 and write profiles/<round>_issue_model.json")
    a = ap.parse_args()
    if a.round:
        import csv
        import json
        sys.path.insert(0, ROOT)
        import bench
        out = {"what": "tools/issue_model.py: VALU issue cycles of the hot kernel priced with the two issue rates of "
                       "gfx950 (profiles/r3_valu_ops.txt) against the SIMD-cycles its launch had",
               "cycles_full_rate": C_FULL, "cycles_sgpr_operand": C_SGPR, "cycles_half_rate": C_HALF,
               "cycles_trip_valu": C_TRIP_VALU, "cycles_lds": C_LDS, "cycles_vmem": C_VMEM, "kernel_source_sha256": bench.kernel_source_hash(),
               "workloads": {}}
        for wl, kern, T in (("c2", [10, 11, 3, 4], 300.0), ("peaks", [10, 10, 3, 4], 600.0), ("c5", [10, 12, 4, 1], None)):
            pj = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (a.round, wl))
            cs = os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.csv" % (a.round, wl))
            if not (os.path.exists(pj) and os.path.exists(cs)):
                continue
            pmc = json.load(open(pj))
            ns = [float(r["AverageNs"]) for r in csv.DictReader(open(cs)) if "k_gram_bitslice" in r["Name"]][0]
            if T is None:   # ragged: the mean column length a wave sweeps = comparisons-weighted; bench's own problem
                args = bench.parse_args(["--workload", wl])
                lens = [len(x) for x in bench.make_problem(args)]
                import numpy as np
                ln = np.array(lens, dtype=np.float64)
                # tiles hold consecutive rows; a tile visits columns 0..its last row: weight of column j ~ rows above it
                wgt = (len(ln) - np.arange(len(ln)))
                T = float((ln * wgt).sum() / wgt.sum())
            m = analyse(a.object, kern, a.shifts_per_block)
            m["measured"] = from_pmc(m, pmc, T, ns)
            m["measured"]["kernel_ms"] = ns / 1e6
            m["mean_column_length"] = T
            out["workloads"][wl] = m
            d = m["measured"]
            print("%-6s trip %d F + %d S + %d H = %.0f cycles, shift %.0f F + %.0f S + %.0f H = %.1f cycles; %.3g shifts, %.3g trips "
                  "(%.0f %% of the VALU instructions, %.0f %% of the issue cycles) -> issue cycles / SIMD-cycles = %.3f at %.2f GHz, "
                  "kernel %.1f ms; with LDS instructions priced (round 5): %.3f, trips %.0f %% of it"
                  % (wl, m["trip"]["full_rate"], m["trip"]["sgpr_operand"], m["trip"]["half_rate"], m["trip"]["cycles"],
                     m["per_shift"]["full_rate"], m["per_shift"]["sgpr_operand"], m["per_shift"]["half_rate"], m["per_shift"]["cycles"],
                     d["shifts"], d["trips"], 100 * d["valu_in_trips"], 100 * d["issue_cycles_in_trips"], d["issue_frac"],
                     d["clock_GHz"], ns / 1e6, d["issue_frac_all"], 100 * d["issue_cycles_all_in_trips"]))
        json.dump(out, open(os.path.join(ROOT, "profiles", "%s_issue_model.json" % a.round), "w"), indent=1, sort_keys=True)
        return 0
    m = analyse(a.object, [int(x) for x in a.kernel.split(",")], a.shifts_per_block)
    t, r = m["trip"], m["per_shift"]
    print("%s" % m["symbol"])
    print("  %d instructions, %d trip copies" % (m["instructions"], m["trip_copies"]))
    print("  one trip:            %3d full-rate + %3d with an SGPR source + %3d half-rate VALU = %5.0f cycles (%d LDS, %d "
          "vector-memory, %d scalar ALU)" % (t["full_rate"], t["sgpr_operand"], t["half_rate"], t["cycles"], t["lds"], t["vmem"], t["salu"]))
    print("  counting loop:       %.1f full-rate + %.1f with an SGPR source + %.1f half-rate VALU = %.1f cycles per shift (%s)"
          % (r["full_rate"], r["sgpr_operand"], r["half_rate"], r["cycles"], r["note"]))
    if a.waves and a.T and a.trips is not None:
        shifts = a.waves * 2 * a.T
        cyc = shifts * r["cycles"] + a.trips * t["cycles"]
        ms = cyc / 1024 / a.clock * 1e3
        print("  predicted: %.3g shifts x %.1f + %.3g trips x %.0f = %.3g SIMD-cycles -> %.1f ms at %.2f GHz on 1024 SIMDs%s"
              % (shifts, r["cycles"], a.trips, t["cycles"], cyc, ms, a.clock / 1e9,
                 "" if a.ms is None else "; measured %.1f ms -> the VALU issue model explains %.0f %% of it" % (a.ms, 100 * ms / a.ms)))


if __name__ == "__main__":
    sys.exit(main())
