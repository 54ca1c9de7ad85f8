#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection CSVs) per kernel: MFMA busy, wait / issue split, LDS bank
conflicts, HBM-side fetch / write bytes per launch (gfx950 corrections of MI355X_MICROARCH.md applied and stated).
usage: pmc_summary.py <pass1_dir> <fetch_dir> <write_dir>"""
import collections
import csv
import glob
import sys


def load(d):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            n[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, {k: len(v) for k, v in n.items()}, dur


def main():
    p1, n1, d1 = load(sys.argv[1])
    pf, nf, _ = load(sys.argv[2])
    pw, nw, _ = load(sys.argv[3])
    keys = sorted((k for k in p1 if "conv_" in k), key=lambda k: -d1[k])
    print("pass 1 (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
    print("        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE), aggregated over every launch of the kernel template:")
    print("  MFMA busy %% = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch duration x 2.4 GHz): the share of the chip's matrix-pipe cycles")
    print("  AT THE PEAK CLOCK the launches kept busy, so busy %% x 157.3 = the fp32 TFLOP/s the MFMAs executed (one v_mfma_f32_32x32x2_f32")
    print("  = 64 busy cycles = 4096 flop -> 'executed TF/s' = busy cycles x 64 / duration).  Round 1 divided by GRBM_GUI_ACTIVE / 8")
    print("  instead; that counter reads high on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back) -- every launch")
    print("  here is 10-50 us -- which made busy %% read ~1.5x too low against the measured flop rate.  It is kept as 'gui-norm'.")
    for k in keys:
        c = p1[k]
        gui = c["GRBM_GUI_ACTIVE"] / 8.0
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        dur_s = d1[k] * 1e-6
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]
        print("  %-44s n=%5d  avg %6.1f us  MFMA busy %5.1f %% (executed %6.1f TF/s; gui-norm %5.1f %%)  of wave cycles: wait %.2f  issue-stall %.2f  active %.2f   LDS conflict/active %.3f"
              % (k, n1[k], d1[k] / n1[k], 100.0 * busy / max(1024.0 * dur_s * 2.4e9, 1.0), busy * 64.0 / max(dur_s, 1e-12) / 1e12,
                 100.0 * busy / max(gui * 1024.0, 1.0), c["SQ_WAIT_ANY"] / wc,
                 c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)))
    print("pass 2 / 3 (FETCH_SIZE, WRITE_SIZE; KB per dispatch summed over the L2 channels):")
    print("  gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> x2 (upper bound for narrow reads); WRITE_SIZE exact")
    for k in keys:
        if k in pf and k in pw:
            f = pf[k]["FETCH_SIZE"] / nf[k] / 1024.0
            w = pw[k]["WRITE_SIZE"] / nw[k] / 1024.0
            print("  %-44s per launch: FETCH_SIZE %8.2f MB (x2 = %8.2f MB)   WRITE_SIZE %7.2f MB" % (k, f, 2 * f, w))


if __name__ == "__main__":
    main()
