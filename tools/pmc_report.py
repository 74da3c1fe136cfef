#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes over `bench.py` into profiles/rNN_mfma_busy.json / rNN_hbm_traffic.json.
usage: pmc_report.py <passA_dir> <passB_dir> <fetch_dir> <write_dir> <out_prefix>
  pass A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
  pass B: SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES
MFMA-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs), duration from the same pass."""
import collections, csv, glob, json, re, sys

CLOCK_HZ, SIMDS = 2.4e9, 1024


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            m = re.search(r"(k_(?:conv|band)_[a-z_]+(?:<[^>]*>)?)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[k]["_ns_" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return per


def mean(v):
    return sum(v) / len(v) if v else None


a, b, f, w = (load(p) for p in sys.argv[1:5])
prefix = sys.argv[5]
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                   # (torch-free import) csrc_sha16: the kernel sources these counters describe; bench.profiled() refuses any other
SHA = sys.argv[6] if len(sys.argv) > 6 else bench.csrc_sha16()
busy = {"note": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes) over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                "--no-aux` (KAN-VGG11, bs 256: every conv-kernel launch of 3 steps); mean per launch of each template instance.  "
                "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs); valu_per_mfma = (SQ_INSTS_VALU - "
                "SQ_INSTS_MFMA) / SQ_INSTS_MFMA (wave instructions); wait_any / wait_inst / active = share of SQ_WAVE_CYCLES-like wave "
                "time parked at s_waitcnt or a barrier / stalled at issue / issuing (pass B counters over their sum).",
        "csrc_sha16": SHA, "instances": {}, "kernels": {}}
fam_acc = collections.defaultdict(lambda: [0.0, 0.0])
for k in sorted(a):
    ca, cb = a[k], b.get(k, {})
    ns = mean(ca["_ns_SQ_VALU_MFMA_BUSY_CYCLES"])
    mf = mean(ca["SQ_VALU_MFMA_BUSY_CYCLES"])
    ent = {"launches": len(ca["SQ_VALU_MFMA_BUSY_CYCLES"]), "avg_launch_us_under_pmc": round(ns / 1e3, 1),
           "mfma_busy": round(mf / (ns * 1e-9 * CLOCK_HZ * SIMDS), 4),
           "valu_per_mfma": round((mean(ca["SQ_INSTS_VALU"]) - mean(ca["SQ_INSTS_MFMA"])) / mean(ca["SQ_INSTS_MFMA"]), 3),
           "lds_per_mfma": round(mean(ca["SQ_INSTS_LDS"]) / mean(ca["SQ_INSTS_MFMA"]), 3),
           "salu_per_mfma": round(mean(ca["SQ_INSTS_SALU"]) / mean(ca["SQ_INSTS_MFMA"]), 3),
           "vmem_per_mfma": round(mean(ca["SQ_INSTS_VMEM"]) / mean(ca["SQ_INSTS_MFMA"]), 3)}
    if cb:
        tot = mean(cb["SQ_WAIT_ANY"]) + mean(cb["SQ_WAIT_INST_ANY"]) + mean(cb["SQ_ACTIVE_INST_ANY"])
        ent.update(wait_any=round(mean(cb["SQ_WAIT_ANY"]) / tot, 3), wait_inst=round(mean(cb["SQ_WAIT_INST_ANY"]) / tot, 3),
                   active=round(mean(cb["SQ_ACTIVE_INST_ANY"]) / tot, 3),
                   lds_bank_conflict_share=round(mean(cb["SQ_LDS_BANK_CONFLICT"]) / max(1.0, mean(cb["SQ_LDS_IDX_ACTIVE"])), 3),
                   mfma_valu_coexec_cycles=round(mean(cb["SQ_VALU_MFMA_COEXEC_CYCLES"])))
    busy["instances"][k] = ent
    fam = re.sub(r"<.*", "", k)
    fam_acc[fam][0] += mf * ent["launches"]
    fam_acc[fam][1] += ns * ent["launches"]
for fam, (mf, ns) in fam_acc.items():
    busy["kernels"][fam] = round(mf / (ns * 1e-9 * CLOCK_HZ * SIMDS), 4)       # time-weighted over the family's launches
json.dump(busy, open(prefix + "_mfma_busy.json", "w"), indent=1)

traffic = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over the same command, mean per launch over all "
                   "launches of the kernel family (all 8 KAN-VGG11 layers, bs 256).  Correction per MI355X_MICROARCH.md section HBM: bytes = "
                   "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE under-reports wide streaming reads by 2x on gfx950; the 4-byte gathers are "
                   "uncalibrated, so this is an upper estimate of the read side; Infinity-Cache hits are counted).", "csrc_sha16": SHA, "kernels": {}}
famf, famw = collections.defaultdict(list), collections.defaultdict(list)
for k, c in f.items():
    famf[re.sub(r"<.*", "", k)] += c["FETCH_SIZE"]
for k, c in w.items():
    famw[re.sub(r"<.*", "", k)] += c["WRITE_SIZE"]
for fam in sorted(famf):
    fk, wk = mean(famf[fam]), mean(famw[fam]) or 0.0
    traffic["kernels"][fam] = {"FETCH_SIZE_KB_per_launch": fk, "launches_profiled": len(famf[fam]), "WRITE_SIZE_KB_per_launch": wk,
                               "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024}
json.dump(traffic, open(prefix + "_hbm_traffic.json", "w"), indent=1)
print(json.dumps(busy["instances"], indent=1))
print(json.dumps(traffic["kernels"], indent=1))
