#!/usr/bin/env python3
"""Turns the SQ / TCC PMC passes of profiles/collect_pmc.sh into profiles/issue.json: per kernel, how busy the vector
ALUs were (the secondary roofline of the tile rasterizer, which is not HBM-bound by construction), the resident waves,
the LDS bank-conflict rate and the memory-side atomic traffic.

    python profiles/pmc_issue.py gpurun_out/<tag> > profiles/issue.json

Units (/opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots" / "Per-instruction cycle constants"):
  SQ_BUSY_CYCLES          cycles, summed over the 32 shader engines  -> kernel cycles = SQ_BUSY_CYCLES / 32
  SQ_ACTIVE_INST_VALU,
  SQ_WAVE_CYCLES, SQ_WAIT_*  quad-cycles (x4 = cycles), summed over all waves / SIMDs (256 CUs x 4 SIMDs)
  SQ_INSTS_VALU           wave-instructions
  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   extra LDS-array cycles / all LDS-array cycles
  TCC_EA0_ATOMIC_sum      64-B atomic requests leaving L2 for the memory side
Three readings of the vector-ALU issue roofline per kernel:
  valu_active           = 4 * SQ_ACTIVE_INST_VALU / (kernel cycles * 1024 SIMDs): the share of every SIMD's time with a
                          vector instruction in execution.  A BUSY counter: two waves' instructions overlap in the SIMD's
                          pipeline, so a saturated SIMD reads above 1.  It says "saturated", not how far from a roofline
                          (`valu_issue_frac` is that value clipped to 1, kept for older readers of the file).
  valu_issue_frac_2cyc  = SQ_INSTS_VALU * 2 cycles / (kernel cycles * 1024): instructions at the guide's best rate (one
                          wave64 v_fma_f32 per 2 SIMD cycles) -- the hard ceiling; the distance to 1 is instruction mix
                          (compares, selects, transcendentals, lane swaps issue slower) plus idle issue slots.
  valu_issue_frac_mix   = SQ_INSTS_VALU * mean issue cost of the kernel's static inner-loop instruction mix (ns per
                          wave-instruction per SIMD, measured per class on this part: profiles/isa_mix.py,
                          profiles/r02_valu_ubench.txt) / 1024 / average launch time: what is left to 1 is idle issue
                          slots (tile tails, staging, waits); a value near 1 says only fewer or cheaper instructions
                          make the kernel faster."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix  # noqa: E402

# label -> (source file, demangled template instance of the bench's configuration)
ISA = {"gs_rasterize_bwd": ("rasterize", "rasterize_bwd_kernel<4, true>"),
       "gs_rasterize_fwd": ("rasterize", "rasterize_fwd_kernel<4>")}

SIMDS = 256 * 4
KERNELS = {
    "gs_projection_fwd": "projection_fwd_kernel", "gs_projection_bwd": "projection_bwd_kernel",
    "gs_sh_bwd_cams": "sh_bwd_cams_kernel",
    "gs_isect_bin_count": "isect_bin_kernel<false, true>", "gs_isect_bin_scatter": "isect_region_scatter_staged_kernel<true>",
    "gs_isect_tile_sort_front": "tile_sort_front_kernel", "gs_rasterize_fwd": "rasterize_fwd_kernel",
    "gs_rasterize_bwd": "rasterize_bwd_kernel",
}


def collect(dirs):
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"]
                    tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[k][row["Counter_Name"]] += 1
    return {k: {c: tot[k][c] / cnt[k][c] for c in tot[k]} for k in tot}


def durations(stats_csv):
    out = {}
    if os.path.exists(stats_csv):
        with open(stats_csv, newline="") as f:
            for row in csv.DictReader(f):
                out[row["Name"]] = float(row["AverageNs"])
    return out


def main():
    root = sys.argv[1]
    per = collect([os.path.join(root, d) for d in ("pmc_sq1", "pmc_sq2", "pmc_tcc")])
    dur = durations(os.path.join(root, "kernel_stats.csv"))
    out = {"source": "rocprofv3 --pmc passes sq1 / sq2 / tcc of profiles/collect_pmc.sh over bench.py (means per launch); "
                     "durations from the --kernel-trace --stats pass of the same command", "kernels": {}}
    for label, sub in KERNELS.items():
        names = [k for k in per if sub in k]
        if not names:
            continue
        k = max(names, key=lambda n: per[n].get("SQ_BUSY_CYCLES", 0.0))  # (the redo launches share the name: mean)
        c = per[k]
        cyc = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
        ns = next((v for n, v in dur.items() if sub in n), None)
        e = {"kernel": k.split("(")[0].replace("void ", ""), "kernel_cycles": round(cyc),
             "avg_launch_ns": ns, "insts_valu": round(c.get("SQ_INSTS_VALU", 0)),
             "insts_valu_trans": round(c.get("SQ_INSTS_VALU_TRANS_F32", 0)),
             "insts_lds": round(c.get("SQ_INSTS_LDS", 0)), "insts_vmem": round(c.get("SQ_INSTS_VMEM", 0)),
             "insts_salu": round(c.get("SQ_INSTS_SALU", 0))}
        if cyc > 0:
            va = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (cyc * SIMDS)
            e["valu_active"] = round(va, 4)
            e["valu_issue_frac"] = round(min(va, 1.0), 4)
            e["valu_issue_frac_2cyc"] = round(c.get("SQ_INSTS_VALU", 0.0) * 2.0 / (cyc * SIMDS), 4)
            if label in ISA and ns:
                try:
                    mix = isa_mix.kernel_mix(*ISA[label])
                    e["valu_mix"] = mix
                    e["vgprs"] = mix["vgprs"]
                    e["valu_issue_frac_mix"] = round(c.get("SQ_INSTS_VALU", 0.0) * mix["mean_issue_ns_per_wave_instruction"]
                                                     / SIMDS / ns, 4)
                except (SystemExit, Exception) as ex:  # (no hipcc where this runs: the two counter readings remain)
                    e["valu_mix_error"] = str(ex)
            e["waves_per_simd"] = round(4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / (cyc * SIMDS), 3)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            e["wave_time_share"] = {"waiting (s_waitcnt / barrier)": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                                    "issue stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                                    "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)}
        la = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        e["lds_bank_conflict_rate"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / la, 4) if la > 0 else 0.0
        at = c.get("TCC_EA0_ATOMIC_sum", 0.0)
        e["atomic_requests_64B"] = round(at)
        if at > 0 and ns:
            e["atomic_GBps"] = round(at * 64.0 / ns, 1)
            e["atomic_frac_of_1300GBps"] = round(at * 64.0 / ns / 1300.0, 4)
        out["kernels"][label] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
