"""Summarise the rocprofv3 --pmc passes of tools/sq_counters.sh: per i2c kernel, counters averaged over its launches (steady
half), per wave, and the fractions of the wave's cycles spent issuing / waiting.
    python tools/sq_summary.py <tag> <dir>   ->   profiles/<tag>_sq_counters.json
SQ cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) count quad-cycles (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import sys


def main():
    tag, d = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "i2c::" not in name:
                continue
            short = name.split("i2c::")[1]
            short = short[: short.index("(")] if "(" in short else short
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "per launch, mean over the steady half of the launches; SQ cycle counters in quad-cycles", "kernels": {}}
    for k, cs in acc.items():
        raw = {c: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for c, v in cs.items()}
        e = {"launches": max(len(v) for v in cs.values()), "raw": raw}
        w, wc = raw.get("SQ_WAVES"), raw.get("SQ_WAVE_CYCLES")
        if w and wc:
            e["per_wave"] = {c.replace("SQ_", "").lower(): raw[c] / w for c in raw if c.startswith("SQ_INSTS")}
            e["per_wave"]["clocks"] = 4 * wc / w
            e["fraction_of_wave_cycles"] = {
                c.replace("SQ_", "").lower(): raw[c] / wc
                for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA",
                          "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC") if c in raw}
        if w and wc and "SQ_INSTS_VALU_MFMA_MOPS_F64" in raw and "SQ_INSTS_VALU" in raw:
            # what the SIMD's fp64 pipe is asked to do: 4 clocks per vector instruction of a wave64, 16 clocks per 512-flop unit
            # of an fp64 matrix instruction (measured: tools/micro/mfma_f64.hip, mfma_f64_4x4.hip), against the wave's clocks
            mfma, mops = raw.get("SQ_INSTS_MFMA", 0.0) / w, raw["SQ_INSTS_VALU_MFMA_MOPS_F64"] / w
            valu = raw["SQ_INSTS_VALU"] / w - mfma
            clocks = 4 * wc / w
            flops_lane = (2 * raw.get("SQ_INSTS_VALU_FMA_F64", 0.0) + raw.get("SQ_INSTS_VALU_ADD_F64", 0.0) + raw.get("SQ_INSTS_VALU_MUL_F64", 0.0)) / w
            e["issue"] = {"vector_insts_per_wave": valu, "mfma_insts_per_wave": mfma, "mfma_512flop_units_per_wave": mops,
                          "fp64_pipe_clocks_per_wave": 4 * valu + 16 * mops, "wave_clocks": clocks,
                          "fp64_issue_frac": (4 * valu + 16 * mops) / clocks, "waves": w, "simds_occupied": min(1.0, w / 1024.0),
                          "fp64_vector_flops_per_lane_per_wave": flops_lane, "mfma_flops_per_wave": 512 * mops}
        out["kernels"][k] = e
    path = f"profiles/{tag}_sq_counters.json"
    json.dump(out, open(path, "w"), indent=1)
    print(path)
    for k, e in out["kernels"].items():
        pw, fr = e.get("per_wave", {}), e.get("fraction_of_wave_cycles", {})
        print(f"{k[:90]:90s} clocks/wave {pw.get('clocks', 0):12.0f} valu {pw.get('insts_valu', 0):9.0f} salu {pw.get('insts_salu', 0):8.0f} "
              f"lds {pw.get('insts_lds', 0):8.0f} | active {fr.get('active_inst_any', 0):.2f} wait {fr.get('wait_any', 0):.2f} "
              f"wait_inst {fr.get('wait_inst_any', 0):.2f} lds_wait {fr.get('wait_inst_lds', 0):.2f}")


if __name__ == "__main__":
    main()
