#!/usr/bin/env python3
"""Instruction-class-weighted issue floor of a straight-line kernel.

  tools/valu_floor.py ISA.s ISSUE_RATES.txt KERNEL_REGEX [--waves-per-row 8] [--col w8]

ISA.s            the gfx950 assembly of the translation unit (hipcc --save-temps)
ISSUE_RATES.txt  the output of tools/ubench/issue_bench on the same box: per instruction class, nanoseconds a SIMD
                 needs per wave-instruction with 1 / 2 / 4 / 8 waves resident (wall clock, so whatever the clock was)
KERNEL_REGEX     selects the kernels (mangled names)

The row-transform kernels have no loops: the static histogram of a kernel IS what one wave executes for one row.  Every
instruction is put into a class the micro-benchmark measured; floor = sum over classes of count x ns, once with the
saturated rates (8 waves per SIMD) and once with the rates at the kernel's own occupancy (4 waves per SIMD: 2 workgroups
of 8 waves per CU).  A row occupies `waves-per-row` waves, i.e. waves-per-row / 4 waves on each SIMD of its CU, and 256
CUs work at once: floor per launch = rows / 256 x (waves-per-row / 4) x floor per wave.
Prints one JSON object per kernel."""
import json
import re
import sys
from collections import Counter


def rates(path, col):
    out = {}
    for ln in open(path):
        m = re.match(r"(.{44})(.*)", ln)
        if not m or "ns/instr/SIMD" not in ln:
            continue
        name = m.group(1).strip()
        cols = dict(re.findall(r"(w\d):.*?=\s*([\d.]+) ns/instr/SIMD", m.group(2)))
        if col in cols:
            out[name] = float(cols[col])
    return out


CLASS = [
    (r"v_mad_u64_u32", "v_mad_u64_u32 (8 independent)", 1.0),
    (r"v_mul_(lo|hi)_u32", "v_mul_hi_u32", 1.0),
    (r"v_(lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|mov_b64)", "v_lshl_add_u64", 1.0),
    # carry pairs were measured as a pair with its wait state: each half costs half the pair
    (r"v_(sub|subb|subrev|subbrev)_co_u32", "v_sub_co + s_nop 1 + v_subb_co (per pair)", 0.5),
    (r"v_(add|addc)_co_u32", "v_add_co + s_nop 1 + v_addc_co (per pair)", 0.5),
    (r"v_cndmask_b32", "csub_select", 1.0),          # derived below: (csub group - sub pair) / 2
    (r"v_add3_u32|v_xad_u32|v_lshl_add_u32|v_add_lshl_u32|v_lshl_or_b32|v_and_or_b32|v_or3_b32|v_bfe_u32|v_perm_b32", "v_add3_u32", 1.0),
    (r"v_alignbit_b32", "v_alignbit_b32", 1.0),
    (r"v_cvt_", "v_cvt_f64_u32", 1.0),
    (r"v_(fma|add|mul|fmac|floor|fract|ldexp|max|min)_f64", "v_fma_f64", 1.0),
    (r"v_mov_b32|v_accvgpr|v_readfirstlane|v_readlane", "v_mov_b32", 1.0),
    (r"v_cmp|v_(add|sub|subrev|and|or|xor|lshlrev|lshrrev|not|min|max|bfi)_[ub]?\w*32|v_and_b32|v_or_b32|v_xor_b32", "v_and_b32", 1.0),
]


def classify(op):
    for pat, cls, w in CLASS:
        if re.match(pat, op):
            return cls, w
    return None, 0.0


def main():
    isa, rate_file, regex = sys.argv[1:4]
    wpr, col = 8, "w8"
    for i, a in enumerate(sys.argv):
        if a == "--waves-per-row":
            wpr = int(sys.argv[i + 1])
        if a == "--col":
            col = sys.argv[i + 1]
    lines = open(isa).read().splitlines()
    tables = {c: rates(rate_file, c) for c in ("w4", "w8")}
    for t in tables.values():
        if "csub: sub, subb, 2 x cndmask (per group)" in t:
            t["csub_select"] = max(0.0, (t["csub: sub, subb, 2 x cndmask (per group)"] - t["v_sub_co + s_nop 1 + v_subb_co (per pair)"]) / 2)
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and re.search(regex, l)]
    dyn = {}      # --dynamic NAME_SUBSTRING:WORKGROUPS:SQ_INSTS_VALU_PER_DISPATCH:IN_SITU_US  (repeatable)
    for i, a in enumerate(sys.argv):
        if a == "--dynamic":
            k, w, v, us = sys.argv[i + 1].rsplit(":", 3)
            dyn[k] = (int(w), float(v), float(us))
    for st in starts:
        name = lines[st].split(":")[0]
        # Round 5: a row kernel holds BOTH arithmetics behind one workgroup-uniform branch (ntt_body: PrimeDev::proth),
        # each path ending in its own s_endpgm.  The histogram is taken over the path the benchmark's rows run: the
        # segment with the complemented quotient digits (v_not_b32) of the Proth-form butterflies, when there is one.
        stop = next((i for i in range(st + 1, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
        ends = [i for i in range(st, stop) if "s_endpgm" in lines[i]]
        segs, a0 = [], st + 1
        for e in ends:
            segs.append((a0, e))
            a0 = e + 1
        def seg_ops(a, b):
            return [l.split()[0] for l in lines[a:b] if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        nots = [sum(1 for o in seg_ops(a, b) if o.startswith("v_not_b32")) for a, b in segs]
        pick = max(range(len(segs)), key=lambda i: nots[i]) if segs and max(nots) > 100 else 0
        ops = seg_ops(*segs[pick]) if segs else []
        hist = Counter(ops)
        valu = {k: v for k, v in hist.items() if k.startswith("v_")}
        by_class, unknown = Counter(), Counter()
        for op, n in valu.items():
            cls, w = classify(op)
            if cls is None:
                unknown[op] += n
            else:
                by_class[(cls, w)] += n
        out = {"kernel": name, "instructions": len(ops), "valu": sum(valu.values()),
               "lds": sum(v for k, v in hist.items() if k.startswith("ds_")),
               "vmem": sum(v for k, v in hist.items() if k.startswith(("buffer_", "global_", "flat_", "scratch_"))),
               "salu": sum(v for k, v in hist.items() if k.startswith("s_") and k not in ("s_nop", "s_waitcnt")),
               "s_nop": hist.get("s_nop", 0), "s_waitcnt": hist.get("s_waitcnt", 0),
               "valu_by_class": {c: n for (c, w), n in by_class.most_common()}, "unclassified_valu": dict(unknown)}
        for c, t in tables.items():
            ns = sum(n * w * t.get(cls, 0.0) for (cls, w), n in by_class.items())
            # the unclassified ones at the simple-op rate
            ns += sum(unknown.values()) * t.get("v_and_b32", 0.0)
            out[f"floor_ns_per_wave_{c}"] = round(ns, 1)
            out[f"floor_ns_per_row_per_cu_{c}"] = round(ns * wpr / 4.0, 1)
            out[f"floor_us_per_1000_rows_chip_{c}"] = round(ns * wpr / 4.0 * 1000 / 256 / 1e3, 3)
        out["path"] = ("proth (segment %d of %d, %d v_not_b32)" % (pick + 1, len(segs), nots[pick])) if segs and max(nots) > 100 else "only"
        for k, (wg, v, us) in dyn.items():
            if k in name:
                per_wave = v / (wg * wpr)
                scale = per_wave / max(1, out["valu"])
                fl = out["floor_ns_per_wave_w8"] * scale * wpr / 4.0 * wg / 256 / 1e3
                out.update({"in_situ_launch": {"workgroups": wg, "avg_us": us},
                            "dynamic_valu_per_wave_SQ_INSTS_VALU": round(per_wave, 1), "dynamic_over_static": round(scale, 4),
                            "floor_us_of_the_in_situ_launch_w8_dynamic_count": round(fl, 1),
                            "measured_over_floor_dynamic_count_w8": round(us / fl, 3)})
        print(json.dumps(out))


if __name__ == "__main__":
    main()
