"""profiles/rNN_pmc.json from the PMC passes of tools/profile_pmc.sh: per-launch means of SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES and
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE for the GEMM and attention kernels.  Usage: python tools/pmc_summary.py gpurun_out r02"""
import collections, csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
out = {"note": "rocprofv3 --kernel-trace --pmc passes of `python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0` (tools/profile_pmc.sh; separate "
               "passes, kernel-trace only), one MI355X.  Per-launch means over all launches of a kernel (sums over the device's SE / XCD "
               "instances as rocprofv3 reports them).  gemm_bf16_persist<EPI, FOLD, DIRECT, MAPPED, KEEP>: <1, true, true, ..> = vision c_fc, "
               "<0, true, false, ..> = QKV, <3, false, true, ..> = out-proj / c_proj.", "kernels": {}}
for sub in ("pmc_mfma", "pmc_lds"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
            if "gemm_bf16" in k or "attn_fwd" in k or "rowstats" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            ent = out["kernels"].setdefault(k, {})
            for c, x in v.items():
                ent[c] = round(sum(x) / len(x))
                ent["launches"] = len(x)
for k, e in out["kernels"].items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CYCLES"):
        e["mfma_busy_per_sq_busy"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"], 2)
    if "SQ_LDS_IDX_ACTIVE" in e and e["SQ_LDS_IDX_ACTIVE"]:
        e["lds_conflict_share"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"], 4)
json.dump(out, open(os.path.join("profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1500])
