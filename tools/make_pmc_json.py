#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/gpu/r3_profile.sh to profiles-ready files:

  <out>/<round>_pmc.json       what bench.py's roofline reads (keyed by the hash of the kernel sources), incl. the VALU
                               instruction CLASS split of the batched scan and the measured issue cost of every class
  <out>/<round>_pmc_<leg>.txt  the per-kernel counter means of every pass, human readable
  <out>/<round>_valu_classes.txt  the derivation of the mix-weighted issue peak, as a table

usage: make_pmc_json.py <pmc dir with knn/ sg/ scan1/ [ubench/] sub-directories> <out dir> [round tag, default r03]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash)

LEGS = {
    # leg -> (kernel-name prefix of the dominant kernel, json key, workload)
    "knn": ("knn_scan_ht", "knn_scan", {"persons": 1_000_000, "places": 100_000, "batch": 16_384, "k": 50}),
    "sg": ("sg_sweep", "sg_sweep", None),
    "scan1": ("knn_scan1", "knn_scan1", {"persons": 1_000_000, "places": 100_000, "k": 50}),
}


def reduce_leg(d):
    """{kernel: {counter: mean per dispatch}}, {kernel: dispatches} over every pass directory of a leg."""
    means, counts = collections.defaultdict(dict), {}
    for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = collections.defaultdict(set)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            counts[k] = len(seen[k])
            for c, x in v.items():
                means[k][c] = x / len(seen[k])
    return means, counts


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:80]


CLASS_COUNTERS = ["INT32", "INT64", "CVT", "MUL_F16", "FMA_F16", "ADD_F16", "TRANS_F16", "ADD_F32", "MUL_F32", "FMA_F32",
                  "TRANS_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64"]


def ubench_table(out, tag):
    """{opcode: cycles per wave64 instruction per SIMD at 8 waves per SIMD} from <tag>_ubench_valu.log."""
    tab = {}
    try:
        for line in open(os.path.join(out, f"{tag}_ubench_valu.log")):
            f = line.split()
            if len(f) == 5 and f[0] == "CLASS" and f[2] == "8":
                tab[f[1]] = float(f[4])
    except OSError:
        pass
    return tab


def valu_mix(means_knn, kernel, ub_means, cycles, fh, image=None):
    """The VALU instructions of one launch split by the SQ's class counters, each class priced with the issue cost the
    microbenchmark measured for the opcodes of that class the scan uses; -> dict for the json.
    Which counter an opcode lands in is read off the same counters on the microbenchmark's own kernels."""
    m = means_knn[kernel]
    total = m.get("SQ_INSTS_VALU")
    if not total or not cycles:
        return None
    # opcode -> class, measured: the class counter that counts (nearly) every instruction of that ubench kernel
    landed = {}
    for k, v in ub_means.items():
        name = short(k)
        tot = v.get("SQ_INSTS_VALU")
        if not name.startswith("k_") or not tot:
            continue
        landed[name] = {c: round(v.get("SQ_INSTS_VALU_" + c, 0.0) / tot, 3) for c in CLASS_COUNTERS
                        if v.get("SQ_INSTS_VALU_" + c, 0.0) > 0.02 * tot}
    # the opcodes of each class that knn_scan_ht issues (csrc/knn_ht.h; static census of its ISA in DESIGN.md section 3)
    # and the issue cost taken for the class: the scan's integer work is v_pk_mad_u16 / SDWA adds / v_alignbit /
    # v_pk_add_u16 (all in the ~4.6-cycle group), its converts are the SDWA v_cvt_f16_u16, MUL_F16 = v_pk_mul_f16,
    # FMA_F16 = v_dot2c_f32_f16; F32 is v_rsq / a few fma; F64 the exact similarity of survivors.  "other" = what
    # no class counter counts (v_mov, v_cmp, v_cndmask, v_readlane ...): priced at the plain 32-bit rate.
    pick = lambda *ops: ([cycles[o] for o in ops if o in cycles] or [4.0])[0]   # the class's dominant opcode comes first
    cost = {"INT32": pick("v_pk_mad_u16", "v_add_u32_sdwa", "v_alignbit_b32", "v_pk_add_u16"),
            "INT64": pick("v_lshl_add_u64"), "CVT": pick("v_cvt_f16_u16_sdwa"), "MUL_F16": pick("v_pk_mul_f16"),
            "FMA_F16": pick("v_dot2c_f32_f16"), "ADD_F16": pick("v_pk_mul_f16"), "TRANS_F16": pick("v_rsq_f32"),
            "ADD_F32": pick("v_fma_f32"), "MUL_F32": pick("v_fma_f32"), "FMA_F32": pick("v_fma_f32"),
            "TRANS_F32": pick("v_rsq_f32"), "ADD_F64": pick("v_fma_f64"), "MUL_F64": pick("v_fma_f64"),
            "FMA_F64": pick("v_fma_f64"), "TRANS_F64": 4 * pick("v_fma_f64")}
    plain = pick("v_and_b32", "v_add_u32")
    counts = {c: m.get("SQ_INSTS_VALU_" + c, 0.0) for c in CLASS_COUNTERS}
    other = max(0.0, total - sum(counts.values()))
    # No class counter counts v_pk_mad_u16, v_pk_add_u16, v_alignbit, v_and, v_mov or v_readlane (see "lands in" below),
    # so the scan's multiply-adds sit in "other".  Their number is exact from the image: every 32-bit word of the head
    # rows (padding included - the kernel multiplies it too) costs a tile 8 v_pk_mad_u16, i.e. tiles x words / 8 wave
    # instructions per launch; one v_alignbit per bound evaluation = the MUL_F16 count (one v_pk_mul_f16 each).
    extra = {}
    if image and image.get("head_words"):
        extra["PK_MAD_U16"] = (min(other, image["tiles"] * image["head_words"] / 8.0), pick("v_pk_mad_u16"))
        extra["ALIGNBIT"] = (min(other - extra["PK_MAD_U16"][0], counts.get("MUL_F16", 0.0)), pick("v_alignbit_b32"))
        other -= extra["PK_MAD_U16"][0] + extra["ALIGNBIT"][0]
    busy = sum(counts[c] * cost[c] for c in CLASS_COUNTERS) + other * plain + sum(n * cy for n, cy in extra.values())
    fh.write(f"VALU instruction classes of {short(kernel)}, one launch (SQ_INSTS_VALU = {total:.0f}):\n")
    fh.write(f"  {'class':12s} {'instructions':>16s} {'share':>7s} {'cycles/instr':>13s}   SIMD-cycles\n")
    for c in CLASS_COUNTERS + list(extra) + ["other"]:
        n, cy = (other, plain) if c == "other" else extra[c] if c in extra else (counts[c], cost[c])
        if n > 0:
            fh.write(f"  {c:12s} {n:16.0f} {n / total:7.3f} {cy:13.3f} {n * cy:13.0f}\n")
    fh.write(f"  mix-weighted cycles per instruction: {busy / total:.3f}  (all at 4 cycles: 4.000, all at 2: 2.000)\n")
    fh.write("which class counter each microbenchmark opcode lands in (share of its SQ_INSTS_VALU):\n")
    for k in sorted(landed):
        fh.write(f"  {k:22s} {landed[k]}\n")
    fh.write("issue cost per opcode, cycles per wave64 instruction per SIMD at 8 waves per SIMD (tools/ubench_valu.hip):\n")
    for k in sorted(cycles):
        fh.write(f"  {k:32s} {cycles[k]:.3f}\n")
    return {"counts": {**{c: counts[c] for c in CLASS_COUNTERS if counts[c] > 0}, **{c: v[0] for c, v in extra.items()}, "other": other},
            "cycles_per_instruction": {**{c: cost[c] for c in CLASS_COUNTERS if counts[c] > 0}, **{c: v[1] for c, v in extra.items()},
                                       "other": plain},
            "image": image,
            "simd_cycles_per_launch": busy, "mix_cycles_per_instruction": busy / total,
            "opcode_cycles": cycles, "opcode_lands_in": landed}


def main():
    src, out = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
    os.makedirs(out, exist_ok=True)
    rec = {"source_hash": bench.kernel_source_hash(), "hash_covers": list(bench.PMC_SOURCES),
           "how": "rocprofv3 --pmc, one pass per counter set (tools/gpu/r3_profile.sh); means per dispatch"}
    cycles = ubench_table(out, tag)
    ub_means = reduce_leg(os.path.join(src, "ubench"))[0] if os.path.isdir(os.path.join(src, "ubench")) else {}
    for leg, (prefix, key, workload) in LEGS.items():
        d = os.path.join(src, leg)
        if not os.path.isdir(d):
            continue
        means, counts = reduce_leg(d)
        with open(os.path.join(out, f"{tag}_pmc_{leg}.txt"), "w") as fh:
            fh.write(f"rocprofv3 --pmc passes, leg '{leg}', kernel sources {rec['source_hash']}; means per dispatch\n")
            own = [k for k in means if "rocprim" not in k and "hipcub" not in k]   # (library sorts of the index build: omitted)
            for k in sorted(own, key=lambda k: -means[k].get("GRBM_GUI_ACTIVE", 0))[:14]:
                fh.write(f"\n{short(k)}  x{counts[k]}\n")
                for c in sorted(means[k]):
                    fh.write(f"    {c:28s} {means[k][c]:18.0f}\n")
        hit = [k for k in means if short(k).startswith(prefix)]
        if not hit:
            continue
        k = max(hit, key=lambda k: means[k].get("GRBM_GUI_ACTIVE", 0))
        m = means[k]
        if workload is None:
            with open(os.path.join(d, "workload.json")) as fh:
                workload = json.load(fh)
        rec[key] = {"kernel": short(k), "dispatches": counts[k], "workload": workload,
                    "insts_valu": m.get("SQ_INSTS_VALU"), "insts_salu": m.get("SQ_INSTS_SALU"),
                    "insts_lds": m.get("SQ_INSTS_LDS"), "fetch_kib": m.get("FETCH_SIZE"),
                    "write_kib": m.get("WRITE_SIZE"), "gui_active_cycles": m.get("GRBM_GUI_ACTIVE"),
                    "tcc_hit": m.get("TCC_HIT_sum"), "tcc_miss": m.get("TCC_MISS_sum")}
        if leg == "knn":
            with open(os.path.join(out, f"{tag}_valu_classes.txt"), "w") as fh:
                image = None
                try:
                    with open(os.path.join(d, "image.json")) as ih:
                        image = json.load(ih)
                except OSError:
                    pass
                mix = valu_mix(means, k, ub_means, cycles, fh, image)
            if mix:
                rec[key]["valu_mix"] = mix
    with open(os.path.join(out, f"{tag}_pmc.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
