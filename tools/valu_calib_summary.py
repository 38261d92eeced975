#!/usr/bin/env python3
"""Distil tools/microbench/valu_calib's output into the constants the roofline accounting uses.

  python tools/valu_calib_summary.py gpurun_out/r03a/valu_calib.txt profiles/r03_valu_calib.json

Per instruction class the SUSTAINED cost in SIMD cycles per wave64 instruction = the host-clock figure at the occupancy where it is
lowest (8 waves per SIMD for every class measured).  Classes: "plain" (v_fma_f32 / v_mul_f32: the full-rate VALU), "double_pass"
(v_pk_fma_f32, DPP-modified adds, v_fma_f64: two passes), "trans" (v_exp_f32 / v_rcp_f32 / v_permlane32_swap: quarter rate), and
"single_wave" = cycles between two independent instructions of ONE wave (what a SIMD with a single resident wave sustains).
tools/pmc_summary.py and bench.py read the result; SQ_INSTS_VALU x "plain" (+ SQ_INSTS_VALU_TRANS x ("trans" - "plain")) / cycles
is then a LOWER bound of the vector pipe's busy fraction (packed / DPP / f64 instructions are not separable in the counters)."""
import json
import sys


def main(src, dst):
    line = next(l for l in open(src) if l.startswith("JSON "))
    d = json.loads(line[5:])
    best = {}
    single = {}
    for r in d["calib"]:
        op = r["op"]
        best[op] = min(best.get(op, 1e9), r["simd_cycles_per_instr_host_clock"])
        if r["waves_per_simd"] == 1:
            single[op] = r["cycles_per_instr_per_wave"]
    out = {"source": src, "device": d["device"], "copy_GBps_float4": d["copy_GBps"],
           "simd_cycles_per_wave_instr": {k: round(v, 3) for k, v in best.items()},
           "classes": {"plain": round(max(best["v_fma_f32"], best["v_mul_f32"]), 3),
                       "double_pass": round(max(best["v_pk_fma_f32"], best["v_add_f32 dpp"], best["v_fma_f64"]), 3),
                       "trans": round(max(best["v_exp_f32"], best["v_rcp_f32"], best["v_permlane32_swap"]), 3),
                       "single_wave": round(single["v_fma_f32"], 3)},
           "note": "SIMD cycles per wave64 instruction, sustained (8 waves per SIMD), from s_memtime / HIP-event timing of dependency-free "
                   "instruction streams on MI355X; a wave64 op on the SIMD-32 takes two passes (~2.3 cycles measured incl. issue "
                   "overhead), packed-f32 / DPP / f64 four, transcendentals and lane swaps eight"}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["classes"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
