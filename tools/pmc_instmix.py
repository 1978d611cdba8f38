#!/usr/bin/env python3
"""tools/pmc_instmix.py <tag> <rocprofv3 --pmc output dirs...> -> profiles/<tag>_pmc_instmix.json

Per kernel of the detect() path (mean over its launches in the profiled bench pass of 1024 frames): wave-instruction counts by
unit and by type, FP64 share, VALU-busy and stall fractions, LDS bank-conflict rate, TA busy, and the VALU ISSUE model the
bench's `issue_roofline` uses.  Round 3 replaced the model's prices by MEASURED ones (tools/ubench/valu_rate.hip on MI355X, four waves per
SIMD, profiles/r03_valu_issue_cost.txt): SIMD cycles per wave64 instruction
    2.1   v_add_f32 / v_sub_f32 / v_mul_f32, v_add_u32 / v_sub_u32, v_and / v_or / v_xor, v_lshrrev_b32, v_mov_b32
    4.2   everything else that is not transcendental: v_fma_f32 / v_fmac_f32, min / max / med3, every conversion, v_lshlrev_b32, v_mad_u32_u24,
          v_mul_lo_u32, v_bfe, v_perm, v_cndmask, compares, v_lshl_add_u64, v_mad_u64_u32, packed f32, and FP64 add / mul / fma
    8.2   v_rcp_f32 / v_exp_f32 ...;   16.1  v_rcp_f64
(the first model -- 2 cycles for everything but FP64 -- priced a conversion or a min3 at half its cost).  The counters split VALU into add /
mul / fma / transcendental f32 and f64, conversions, int32 and int64; int32 mixes both classes and is priced at 3.15 (lo / hi bounds with
2.1 / 4.2 are reported as well); what the typed counters do not cover is priced at 4.2.
issue_bound_ms = issue_cycles / (1024 SIMDs * 2.4 GHz).  `valu_busy_simd` is the measured counterpart: SQ_ACTIVE_INST_VALU (units of four
cycles, like SQ_WAVE_CYCLES) against the 1024 SIMDs' cycles of the launch."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, CLOCK_GHZ = 256 * 4, 2.4


def main(tag, dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(float)
            span = {}
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                per[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
                span[(r["Dispatch_Id"], k)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            for (_, k, c), v in per.items():
                acc[k][c].append(v)
            for (_, k), ms in span.items():
                dur[k].append(ms)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import srcsha
    out = {"tag": tag, "sources_sha256": srcsha.sources_sha256(), "git_head": srcsha.git_head(), "workload": "bench.py --frames 1024 --chunk 1024 (one pass of 1024 synthetic 1080p frames), means per launch",
           "method": "rocprofv3 --kernel-trace --pmc, four separate passes of <= 8 SQ counters (tools/pmc_instmix.sh); durations are the "
                     "profiled (counter-collecting) launches' and run slower than unprofiled ones",
           "kernels": {}}
    for k in sorted(acc):
        if not k.startswith("ctag::"):
            continue
        m = {c: sum(v) / len(v) for c, v in acc[k].items()}
        g = lambda c: m.get(c, 0.0)  # noqa: E731
        valu = g("SQ_INSTS_VALU")
        if valu < 1e4:
            continue
        f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")
        fast = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32")
        tr32, tr64, i32 = g("SQ_INSTS_VALU_TRANS_F32"), g("SQ_INSTS_VALU_TRANS_F64"), g("SQ_INSTS_VALU_INT32")
        full = max(valu - fast - tr32 - tr64 - i32, 0.0)  # fma f32, conversions, int64, f64 add / mul / fma, and whatever no typed counter covers
        base = 2.1 * fast + 4.2 * full + 8.2 * tr32 + 16.1 * tr64
        issue_cycles, issue_lo, issue_hi = base + 3.15 * i32, base + 2.1 * i32, base + 4.2 * i32
        to_ms = lambda c: round(c / (SIMDS * CLOCK_GHZ * 1e9) * 1e3, 4)  # noqa: E731
        launch_cycles = (sum(dur[k]) / len(dur[k])) * 1e-3 * CLOCK_GHZ * 1e9
        e = {"launch_ms_profiled": round(sum(dur[k]) / len(dur[k]), 4), "waves": g("SQ_WAVES"),
             "wave_instructions": {"valu": valu, "salu": g("SQ_INSTS_SALU"), "lds": g("SQ_INSTS_LDS"), "vmem_rd": g("SQ_INSTS_VMEM_RD"),
                                   "vmem_wr": g("SQ_INSTS_VMEM_WR")},
             "valu_by_type": {"add_f64": g("SQ_INSTS_VALU_ADD_F64"), "mul_f64": g("SQ_INSTS_VALU_MUL_F64"), "fma_f64": g("SQ_INSTS_VALU_FMA_F64"),
                              "trans_f64": g("SQ_INSTS_VALU_TRANS_F64"), "add_f32": g("SQ_INSTS_VALU_ADD_F32"), "mul_f32": g("SQ_INSTS_VALU_MUL_F32"),
                              "fma_f32": g("SQ_INSTS_VALU_FMA_F32"), "trans_f32": g("SQ_INSTS_VALU_TRANS_F32"), "cvt": g("SQ_INSTS_VALU_CVT"),
                              "int32": g("SQ_INSTS_VALU_INT32"), "int64": g("SQ_INSTS_VALU_INT64")},
             "fp64_share_of_valu": round(f64 / valu, 4),
             "valu_busy": round(g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), 4) if g("SQ_WAVE_CYCLES") else None,
             "wait_inst_any_frac": round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4) if g("SQ_WAVE_CYCLES") else None,
             "wait_any_frac": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4) if g("SQ_WAVE_CYCLES") else None,
             "lds_bank_conflict_rate": round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4) if g("SQ_LDS_IDX_ACTIVE") else None,
             "ta_busy_avr_pct": g("TA_BUSY_avr"), "ta_busy_cycles_sum": g("TA_TA_BUSY_sum"), "grbm_gui_active": g("GRBM_GUI_ACTIVE"),
             "tcp_cache_accesses": g("TCP_TOTAL_CACHE_ACCESSES_sum"), "tcp_tcc_read_req": g("TCP_TCC_READ_REQ_sum"),
             "raw": {c: m[c] for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                                       "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS") if c in m},
             "valu_busy_simd": round(4.0 * g("SQ_ACTIVE_INST_VALU") / (SIMDS * launch_cycles), 4) if launch_cycles else None,
             "issue_model": {"issue_cycles": issue_cycles, "issue_bound_ms": to_ms(issue_cycles), "issue_bound_ms_range": [to_ms(issue_lo), to_ms(issue_hi)]}}
        out["kernels"][k] = e
    path = os.path.join(ROOT, "profiles", tag + "_pmc_instmix.json")
    json.dump(out, open(path, "w"), indent=1)
    print("%-40s %9s %9s %7s %7s %9s %9s" % ("kernel", "ms(prof)", "VALU", "fp64", "busy", "issue ms", "TA busy%"))
    for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["launch_ms_profiled"]):
        print("%-40s %9.3f %9.3g %7.3f %7s %9.3f %9s" % (k[:40], e["launch_ms_profiled"], e["wave_instructions"]["valu"], e["fp64_share_of_valu"],
                                                         e["valu_busy"], e["issue_model"]["issue_bound_ms"], e["ta_busy_avr_pct"]))
    print("written", path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
