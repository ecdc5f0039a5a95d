"""Summarise the CSVs of tools/prof_pmc.sh: per kernel (name substring filter), average counter value per dispatch, and
the derived figures DESIGN.md quotes (MI355X_MICROARCH.md units: SQ_VALU_MFMA_BUSY_CYCLES = 16 per 16x16x32 / 32 per
32x32x16 f16 MFMA, summed over the 1024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs)."""
import csv, glob, os, sys, collections
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")


def short(k):
    return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]


allc = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "pmc*.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if flt in k:
            acc[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(os.path.basename(f), k)
        for c, v in cs.items():
            print(f"    {c:34s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
            allc[k][c] = sum(v) / len(v)
print("derived (per dispatch):")
for k, c in allc.items():
    out = []
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        out.append(f"matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (c['GRBM_GUI_ACTIVE'] / 8) * 100:.1f} % of the kernel's cycles")
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_F16") and "SQ_INSTS_VALU" in c:      # (a kernel without MFMAs - the lookup - has the counter at 0)
        flop = c["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512
        out.append(f"{flop / 1e9:.2f} GFLOP of f16 MFMA; vector instructions (MFMAs included) per 16 384-FLOP MFMA {c['SQ_INSTS_VALU'] / (flop / 16384):.2f}")
    elif "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
        out.append(f"no MFMA; {c['SQ_INSTS_VALU'] / c['SQ_WAVES']:.1f} vector instructions per wave")
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"LDS bank conflicts {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'] * 100:.1f} % of the LDS-active cycles")
    if "TCC_EA0_RDREQ_sum" in c:
        out.append(f"fabric reads {c['TCC_EA0_RDREQ_sum'] * 128 / 1e6:.1f} MB" + (f", written {c['WRITE_SIZE'] * 1024 / 1e6:.1f} MB" if "WRITE_SIZE" in c else ""))
    if out:
        print("  ", k, "\n      " + "; ".join(out))
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        if flt in r["Name"]:
            print("trace", short(r["Name"]), "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
