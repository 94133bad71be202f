"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
  <tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.csv, <tag>_timeline.txt, <tag>_roofline_table.md and - for
the default workload (or --traffic-json NAME) - dw_fwd1_traffic.json, the constants bench.py quotes.
Usage: python tools/summarize_profiles.py <tag> [--traffic-json dw_fwd1_traffic.json] [--title "..."]"""
import argparse
import csv
import json
import os
import re
import shutil

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--traffic-json", default=None)
ap.add_argument("--title", default="128^3 x 4 fp32 training step, all three streams running")
args = ap.parse_args()
tag = args.tag
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "timeline.txt")):
    shutil.copy(os.path.join(src, "timeline.txt"), os.path.join(dst, f"{tag}_timeline.txt"))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", n)
    return m.group(1) if m else n


stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))}
steps = max(int(stats[k]["Calls"]) for k in stats if "adam_kernel" in k) if any("adam_kernel" in k for k in stats) else None
pmc = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(src, "pmc_avg.csv")))}
ours = lambda k: "at::native" not in k and "rocclr" not in k and "Cijk" not in k
CU, SIMD, XCD = 256, 4, 8
rows = []
for k, st in stats.items():
    if not ours(k):
        continue
    p = pmc.get(k, {})
    f = lambda c: float(p[c]) if p.get(c) not in (None, "") else None
    fk, wk = f("FETCH_SIZE"), f("WRITE_SIZE")
    mfma, gui, sqb = f("SQ_VALU_MFMA_BUSY_CYCLES"), f("GRBM_GUI_ACTIVE"), f("SQ_BUSY_CYCLES")
    us = float(st["AverageNs"]) / 1e3
    mb = None if fk is None or wk is None else (2.0 * fk + wk) * 1024 / 1e6
    # gfx94x MfmaUtil formula (ROCm 7.2 ships no gfx950 derived counters): MFMA-busy cycles summed over all SIMDs against
    # the cycles the chip was active (GRBM_GUI_ACTIVE is the sum over the 8 XCDs) x 256 CUs x 4 SIMDs
    util = None if not mfma or not gui else mfma / (gui / XCD * CU * SIMD)
    rows.append(dict(total=float(st["TotalDurationNs"]), name=short(k), calls=int(st["Calls"]), us=us, mb=mb, fk=fk, wk=wk,
                     mfma=mfma, gui=gui, sqb=sqb, util=util))
rows.sort(key=lambda r: -r["total"])
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as f:
    f.write("kernel,calls,avg_us,FETCH_SIZE_kb_avg_raw,WRITE_SIZE_kb_avg,SQ_VALU_MFMA_BUSY_CYCLES_avg,SQ_BUSY_CYCLES_avg,"
            "GRBM_GUI_ACTIVE_avg,note: FETCH_SIZE x2 for 16-B/lane streaming reads on gfx950; GRBM_GUI_ACTIVE = sum over 8 XCDs\n")
    g = lambda v: "" if v is None else f"{v:.1f}"
    for r in rows:
        f.write(f'"{r["name"]}",{r["calls"]},{r["us"]:.2f},{g(r["fk"])},{g(r["wk"])},{g(r["mfma"])},{g(r["sqb"])},{g(r["gui"])},\n')
per = (lambda r: r["calls"] / steps) if steps else (lambda r: float("nan"))
out = [f"# Per-kernel duration, HBM traffic and MFMA-busy ({tag}; {args.title})", "",
       "rocprofv3 kernel-trace average duration; PMC per launch from separate counter-only passes: HBM MB = `FETCH_SIZE` x2 + "
       "`WRITE_SIZE` (gfx950 correction of MI355X_MICROARCH.md), TB/s = that / duration against the 8 TB/s HBM3E peak; "
       "MFMA-busy = `SQ_VALU_MFMA_BUSY_CYCLES` / (`GRBM_GUI_ACTIVE`/8 x 256 CUs x 4 SIMDs) (the gfx94x MfmaUtil formula; "
       "GRBM_GUI_ACTIVE is summed over the 8 XCDs).  Kernels below ~20 us are latency-, not bandwidth-bound.", "",
       "| kernel | launches/step | avg us | us/step | HBM MB/launch | TB/s | of 8 TB/s | MFMA-busy |", "|---|---|---|---|---|---|---|---|"]
tot = 0.0
for r in rows[:40]:
    tbs = None if r["mb"] is None or not r["us"] else r["mb"] / r["us"]
    perstep = r["total"] / steps / 1e3 if steps else float("nan")
    tot += perstep
    out.append(f"| `{r['name']}` | {per(r):.1f} | {r['us']:.1f} | {perstep:.1f} | " +
               ("-" if r["mb"] is None else f"{r['mb']:.1f}") + " | " + ("-" if tbs is None else f"{tbs:.2f}") + " | " +
               ("-" if tbs is None else f"{tbs / 8.0:.2f}") + " | " + ("-" if r["util"] is None else f"{100 * r['util']:.1f} %") + " |")
alltot = sum(r["total"] for r in rows) / steps / 1e3 if steps else float("nan")
out += ["", f"Sum of kernel time per step (all {len(rows)} kernels of the step): {alltot:.1f} us; launches per step: "
        f"{sum(r['calls'] for r in rows) / steps if steps else float('nan'):.1f}; bench line of the same box: "
        f"{bench.get('ms_per_step')} ms/step, {bench.get('value')} {bench.get('unit')}."]
open(os.path.join(dst, f"{tag}_roofline_table.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:20]))

if args.traffic_json:
    # block 1's depthwise forward (the bench line's roofline kernel) + the seven forward depthwise launches of a step, in
    # launch order, from the kernel trace (stride-1 kernels also run as bwd-data launches: the trace tells them apart)
    bf16 = bench.get("dtype") == "bf16"
    canon = "dw_s2_wave_kernel<4,5,4,bf16>" if bf16 else "dw_s2_wave_kernel<4,5,4>"  # the name bench.py quotes
    key = [r for r in rows if r["name"].startswith("dw_s2_wave_kernel<4, 5, 4") and
           (("unsigned short" in r["name"]) == bf16)]
    tr = list(csv.DictReader(open(os.path.join(src, "kernel_trace.csv"))))
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(tr) if "adam_kernel" in r["Kernel_Name"]]
    per_layer = [[] for _ in range(7)]
    isdw = lambda n: re.match(r"dw_(s[12]_wave_kernel|fwd_stream_kernel|fwd_bf16_kernel|fwd_kernel)", short(n)) is not None
    for a, b in zip(ends[5:-1], ends[6:]):
        dws = [r for r in tr[a + 1:b] if isdw(r["Kernel_Name"])][:7]
        if len(dws) == 7:
            for i, r in enumerate(dws):
                per_layer[i].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lay = [sum(v) / len(v) for v in per_layer] if all(per_layer) else None
    esz = 2.0 if bf16 else 4.0
    dims = [64 ** 3, 32 ** 3, 16 ** 3, 16 ** 3, 8 ** 3, 8 ** 3, 4 ** 3, 4 ** 3]
    chans = [32, 64, 128, 128, 256, 256, 512]
    lay_b = [esz * 4 * c * (dims[i] + dims[i + 1]) + 4.0 * c * 27 for i, c in enumerate(chans)]
    if key:
        r = key[0]
        hbm = r["mb"] * 1e6
        agg = None
        if lay:
            gbs = sum(lay_b) / (sum(lay) * 1e-6) / 1e9
            agg = {"sum_launch_us": round(sum(lay), 2), "achieved": round(gbs, 1), "frac": round(gbs / 8000.0, 4),
                   "per_layer_us": [round(u, 2) for u in lay], "profile": f"profiles/{tag}_kernel_stats.csv (kernel trace of the same run)"}
        json.dump({
            "kernel": canon, "round": 2, "profile": tag, "FETCH_SIZE_kb_avg": r["fk"], "WRITE_SIZE_kb_avg": r["wk"],
            "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streaming reads -> x2 "
                          "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
            "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": lay_b[0], "avg_launch_us_rocprof": round(r["us"], 2),
            "depthwise_fwd_all_layers_rocprof": agg,
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py "
                      "--steps 6 --warmup 6 --no-cpu-baseline --no-aggregate; durations from rocprofv3 --kernel-trace --stats "
                      "(tools/collect_profiles.sh)",
        }, open(os.path.join(dst, args.traffic_json), "w"), indent=1)
        print(f"dw_fwd1: {r['us']:.2f} us, traffic {hbm / 1e6:.1f} MB per launch ({hbm / lay_b[0]:.3f} x algorithmic); all layers: {agg}")
