"""Post-process the rocprofv3 outputs of tools/profile_round.sh into the judged artefacts under profiles/:
  <tag>_kernel_stats.csv      the --kernel-trace --stats summary of one `bench.py` run (copied as produced)
  <tag>_roofline.json/.md     per-kernel table: calls, average duration, algorithmic work per launch where DESIGN.md section 4
                              defines it, achieved rate, fraction of the MI355X roofline
  traffic.json                HBM bytes per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes
                              (FETCH_SIZE doubled: gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md HBM section)
Usage: python tools/collect_profiles.py <gpurun_out dir> <tag> [model] [batch]"""
import csv, glob, json, os, shutil, statistics, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, tag = sys.argv[1], sys.argv[2]
model = sys.argv[3] if len(sys.argv) > 3 else "vit-large-patch14-224"
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 256
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return hits[-1] if hits else None


stats = one("prof_stats/**/*kernel_stats.csv")
rows = []
if stats:
    shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))

from openvision_amd import preset
cfg = preset(model)
v, t = cfg["vision_cfg"], cfg["text_cfg"]
g = v["image_size"] // v["patch_size"]
Lv, Dv, Dt, T = g * g + 1, v["width"], t["width"], t["context_length"]
Fv = int(Dv * v["mlp_ratio"])
Mv_main = (batch - 1) * Lv            # the tower peels the last image onto a side stream (tile quantisation)
PEAK_TF, PEAK_GBS = 2500.0, 8000.0
# kernels whose per-launch algorithmic work is unambiguous on this workload (vision tower main part)
known = {
    "gemm_bf16_persist<1, true, false, false, false, false>": ("mfma", 2.0 * Mv_main * Fv * Dv, "vision mlp.c_fc (LN fold + erf-GELU, LDS-transposed whole-line stores: the form large outputs take)"),
    "gemm_bf16_persist<1, true, true, false, false, false>": ("mfma", 2.0 * Mv_main * Fv * Dv, "vision mlp.c_fc (LN fold + erf-GELU, direct-store epilogue)"),
    "gemm_bf16_persist<1, true, true, false, false>": ("mfma", 2.0 * Mv_main * Fv * Dv, "vision mlp.c_fc (LN fold + erf-GELU, direct-store epilogue) [kernel name before the STATS parameter]"),
    "gemm_bf16_persist<1, true, true, false>": ("mfma", 2.0 * Mv_main * Fv * Dv, "vision mlp.c_fc (LN fold + erf-GELU, direct-store epilogue) [kernel name before the KEEP parameter]"),
    "gemm_bf16_persist<1, true, true>": ("mfma", 2.0 * Mv_main * Fv * Dv, "vision mlp.c_fc (LN fold + erf-GELU, direct-store epilogue) [earlier kernel name]"),
    "gemm_bf16_persist<1, true>": ("mfma", 2.0 * Mv_main * Fv * Dv, "vision mlp.c_fc (LN fold + erf-GELU) [round-1 kernel name]"),
    "rowstats_rows<2>": ("hbm", None, "row mean/rstd of the residual stream (both towers, tail images: mixed sizes)"),
}
if any("gemm_bf16_persist<1, true, false, false, false, false>" in r["Name"] for r in rows):
    # the vision c_fc runs in the LDS-transposed form: the direct-form launches in this trace are the text tower's (other M)
    known = {k: v for k, v in known.items() if not k.startswith("gemm_bf16_persist<1, true, true")}
table = []
for r in rows:
    name = r["Name"]
    short = name.replace("void (anonymous namespace)::", "").split("(")[0]
    ent = {"kernel": short, "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
           "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "pct": float(r["Percentage"])}
    if short in known and known[short][1]:
        bound, work, what = known[short]
        rate = work / (float(r["AverageNs"]) * 1e-9)
        ent.update(bound=bound, what=what, work_per_launch=work,
                   achieved=round(rate / (1e12 if bound == "mfma" else 1e9), 1), unit="TFLOP/s" if bound == "mfma" else "GB/s",
                   frac=round(rate / (1e12 * PEAK_TF if bound == "mfma" else 1e9 * PEAK_GBS), 4))
    table.append(ent)
if table:
    json.dump({"model": model, "batch": batch, "kernels": table}, open(os.path.join(out, f"{tag}_roofline.json"), "w"), indent=1)
    with open(os.path.join(out, f"{tag}_roofline.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats, `bench.py` ({model}, batch {batch}) — per kernel\n\n"
                "Class-level achieved rates (all launches of a class, both towers) are in the `breakdown` object of the bench JSON.\n\n"
                "| kernel | calls | avg us | total ms | % | achieved | frac of roofline |\n|---|---|---|---|---|---|---|\n")
        for e in table:
            f.write(f"| `{e['kernel']}` | {e['calls']} | {e['avg_us']} | {e['total_ms']} | {e['pct']} | "
                    f"{str(e.get('achieved', '')) + ' ' + e.get('unit', '') if 'achieved' in e else ''} | {e.get('frac', '')} |\n")

# ---- HBM traffic of the dominant kernel from the two PMC passes ----
def pmc(sub, counter):
    f = one(f"{sub}/**/*counter_collection.csv")
    if not f:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and "gemm_bf16_persist<1, true" in r["Kernel_Name"]]
    return vals


fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
if fe and wr:
    fkb, wkb = statistics.median(fe), statistics.median(wr)
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) on `python3 bench.py --steps 2 "
                       "--warmup 1 --cpu-seconds 0`; per-launch medians over the launches of gemm_bf16_persist<1, true, *>; bytes = "
                       "(2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md)",
               "model": model, "batch": batch, "kernel": "gemm_bf16_persist<1, true, *> (vision mlp.c_fc)", "launches": len(fe),
               "FETCH_SIZE_KB_median": fkb, "WRITE_SIZE_KB_median": wkb,
               "gemm_fc_bytes_per_launch": (2 * fkb + wkb) * 1024.0,
               "algorithmic_bytes_per_launch": 2.0 * (Mv_main * Dv + Fv * Dv + Mv_main * Fv)},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("profiles written:", sorted(os.listdir(out)))
