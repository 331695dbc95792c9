"""Per-kernel table of ONE training step out of a rocprofv3 --kernel-trace of tools/train_step_probe.py: the window between the last
two clip_loss_bwd launches (= the backward of step n-1 and the forward of step n: one step's worth of launches, back to back).
Usage: python tools/train_step_table.py gpurun_out/prof_train/run_kernel_trace.csv [out.md]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:80]


marks = [i for i, r in enumerate(rows) if "clip_loss_bwd<false>" in r["Kernel_Name"]]
i0, i1 = marks[-2], marks[-1]
t0, t1 = int(rows[i0]["Start_Timestamp"]), int(rows[i1]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[i0:i1]:
    k = short(r["Kernel_Name"])
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
lines = [f"window {(t1 - t0) / 1e6:.1f} ms, kernels busy {tot / 1e6:.1f} ms, {i1 - i0} launches", "",
         "| kernel | launches | avg us | ms per step | % |", "|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if v[1] / tot < 0.001:
        continue
    lines.append(f"| `{k}` | {v[0]} | {v[1] / v[0] / 1e3:.1f} | {v[1] / 1e6:.2f} | {100 * v[1] / tot:.1f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
