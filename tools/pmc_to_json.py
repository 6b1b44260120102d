#!/usr/bin/env python3
"""tools/rocpd_pmc.py's summary of the three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU over
`bench.py --steps 2 --warmup 1 --mults-per-step 4 --no-extras --cpu-sample 0`) -> the two JSON records bench.py and DESIGN.md
cite: the HBM traffic of ONE fresh multiply of the batch (sum over its launches) and, per roofline kernel, traffic against
algorithmic bytes.   usage: tools/pmc_to_json.py pmc_summary.txt ROUND > out.json"""
import json
import re
import sys

path, rnd = sys.argv[1], sys.argv[2]
rows = {}
for ln in open(path):
    m = re.match(r"(.{60,70}?)\s+(\d+)\s+(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*(\d*)\s*$", ln)
    if m:
        name, wg, ctr, disp, us, per, byts = m.groups()
        rows.setdefault((name.strip(), int(wg)), {})[ctr] = (int(disp), float(us), float(per), int(byts) if byts else None)

N, B, L = 16384, 128, 16
# the launches of one level-1 fresh multiply of the 128-pair batch: (name prefix, workgroups, launches per multiply)
ONE = [("void hx::ntt_moddown_apply_kernel<14, false>", 8192, 1), ("void hx::ntt_moddown_apply_tensor_kernel<14, false>", 6144, 1),
       ("void hx::keyswitch_kernel<3>", 8206, 1), ("void hx::ntt_row_kernel<14, false, 8>", 6400, 1),
       ("void hx::break_digits_fast_kernel<true, true>", 16384, 1), ("void hx::ntt_row_kernel<14, true, 1>", 2048, 1),
       ("void hx::ntt_moddown_prep_tensor_kernel<14>", 384, 1), ("void hx::ntt_moddown_prep_kernel<14>", 512, 1),
       ("void hx::moddown_S_kernel<0>", 8192, 2), ("void hx::embed_norm_r16_kernel<hx::NormSrcXS>", 512, 1),
       ("void hx::embed_norm_r16_kernel<hx::NormSrcXS>", 384, 1), ("void hx::embed_norm_r16_kernel<hx::NormSrcF64>", 384, 1)]


def find(prefix, wg):
    for (name, w), v in rows.items():
        if w == wg and prefix.startswith(name.rstrip(".").rstrip()[:len(name) - 3]) or (w == wg and name.startswith(prefix[:55])):
            return v
    return None


per, total = {}, 0
for prefix, wg, k in ONE:
    v = find(prefix, wg)
    if not v or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        per[f"{prefix[9:]} ({wg})"] = None
        continue
    f, w = v["FETCH_SIZE"][3], v["WRITE_SIZE"][3]
    per[f"{prefix[9:]} ({wg}) x{k}"] = [f * k, w * k]
    total += (f + w) * k
alg = {"ntt_moddown_apply_kernel<14, false>": (8192, (B * 4) * (L * 16 * N - 8 * N + 16 * N)),
       "ntt_moddown_apply_tensor_kernel<14, false>": (6144, B * 8 * N * (4 * (L - 1) + 3 * L + 6)),
       "ntt_row_kernel<14, false, 8>": (6400, 6400 * 16 * N), "keyswitch_kernel<3>": (8206, None)}
roof = []
for name, (wg, ab) in alg.items():
    v = find("void hx::" + name, wg)
    if not v:
        continue
    f, w = v["FETCH_SIZE"][3], v["WRITE_SIZE"][3]
    r = {"kernel": name, "workgroups": wg, "fetched_bytes_per_launch": f, "written_bytes_per_launch": w,
         "traffic_bytes_per_launch": f + w, "avg_us_in_the_pmc_pass": v["FETCH_SIZE"][1],
         "SQ_INSTS_VALU_per_dispatch": int(v["SQ_INSTS_VALU"][2]) if "SQ_INSTS_VALU" in v else None}
    if ab:
        r["algorithmic_bytes_per_launch"] = ab
        r["traffic_over_algorithmic"] = round((f + w) / ab, 3)
    roof.append(r)
json.dump({"workload": f"BGV m=32768 bits=950 fresh multiplyBy, batch 128, measured noise, round-{rnd} kernels",
           "batch": B, "bits": 950, "traffic_GB_per_multiply_of_the_batch": round(total / 1e9, 2),
           "per_kernel_bytes_fetched_written_per_multiply_of_the_batch": per, "roofline_kernels": roof,
           "note": "sum over every launch of one fresh multiplyBy of the 128-pair batch of 2 x FETCH_SIZE + WRITE_SIZE "
                   "(gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes, /opt/skills/guides/MI355X_MICROARCH.md); "
                   "FETCH_SIZE is counted at the L2: a read served by the Infinity Cache counts, HBM does not see it",
           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, three separate passes over `bench.py --gpus 1 --steps 2 "
                     f"--warmup 1 --mults-per-step 4 --no-extras --cpu-sample 0` (tools/gpu_calls.sh OUT pmc), summarised by tools/rocpd_pmc.py "
                     f"into profiles/r{int(rnd):02d}_pmc_bench_command.txt, this file by tools/pmc_to_json.py"}, sys.stdout, indent=1)
