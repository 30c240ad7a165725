#!/usr/bin/env python3
"""Copy the outputs of scripts/r4/profile_all.sh (gpurun_out/r4prof) into profiles/r04_*: bench lines, rocprofv3 kernel statistics and a
PMC traffic summary with the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md: FETCH_SIZE counts half of the bytes of wide coalesced
streaming reads; WRITE_SIZE is exact)."""
import collections, csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "gpurun_out", "r4prof")
D = os.path.join(ROOT, "profiles")
for f in glob.glob(os.path.join(P, "*.json")):
    b = os.path.basename(f)
    if b.startswith(("bench_10m", "bench_1m", "nonlinear", "assembly", "update_linear")):
        shutil.copy(f, os.path.join(D, "r04_" + b))
for src, dst in (("trace_kernel_stats.csv", "r04_rocprofv3_kernel_stats_cg_ssor_10m.csv"), ("ilu_kernel_stats.csv", "r04_rocprofv3_kernel_stats_bicgstab_ilu0_10m.csv"),
                 ("asm_kernel_stats.csv", "r04_rocprofv3_kernel_stats_assembly_10m.csv")):
    if os.path.exists(os.path.join(P, src)):
        shutil.copy(os.path.join(P, src), os.path.join(D, dst))
if os.path.exists(os.path.join(P, "bench_trace.json")):
    shutil.copy(os.path.join(P, "bench_trace.json"), os.path.join(D, "r04_bench_under_rocprof_trace.json"))

def counters(name):
    agg = collections.defaultdict(list)
    f = os.path.join(P, name)
    if not os.path.exists(f):
        return agg
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg

fetch, write, ilu = counters("pmc_fetch_counters.csv"), counters("pmc_write_counters.csv"), counters("ilu_fetch_counters.csv")
out = ["rocprofv3 PMC passes (separate runs: --pmc FETCH_SIZE, --pmc WRITE_SIZE; scripts/r4/profile_all.sh) of",
       "  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline   (10.125M DOF, CG + multicolour SSOR: Eisenstat's form = the headline, the standard loop = the variant; MI355X)",
       "Counter unit: KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.", ""]
B = {}
for name, agg in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    out.append(name + " per launch (mean over launches)")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:18]:
        m = sum(v) / len(v)
        b = m * 1024 * (2 if name == "FETCH_SIZE" else 1)
        B[(name, k)] = (b, len(v))
        out.append("  %-46s launches=%5d  counter=%12.1f KiB  -> %8.1f MB%s" % (k[:46], len(v), m, b / 1e6, " (x2 corrected)" if name == "FETCH_SIZE" else ""))
    out.append("")

def tot(prefix):
    f = sum(b * n for (nm, k), (b, n) in B.items() if nm == "FETCH_SIZE" and k.startswith(prefix))
    w = sum(b * n for (nm, k), (b, n) in B.items() if nm == "WRITE_SIZE" and k.startswith(prefix))
    n = sum(n for (nm, k), (b, n) in B.items() if nm == "FETCH_SIZE" and k.startswith(prefix))
    return f, w, n

traffic = {"_comment": "HBM traffic per launch from rocprofv3 PMC passes (profiles/r04_pmc_summary.txt): 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, gfx950 correction per MI355X_MICROARCH.md.  k_spmv keyed by block rows, k_assemble_c3d8 by elements (one assembly = its 8 colour launches)."}
f, w, n = tot("void k_spmv<0, 1")
if n:
    t = (f + w) / n
    out.append("k_spmv<0,1,...> per launch: fetch %.0f MB + write %.0f MB = %.0f MB HBM traffic vs algorithmic 7009 MB (SURVEY 8d): %+.1f %%" % (f / n / 1e6, w / n / 1e6, t / 1e6, (t / 7.009e9 - 1) * 100))
    traffic["k_spmv"] = {"3375000": {"traffic_bytes": round(t), "fetch_bytes": round(f / n), "write_bytes": round(w / n), "source": "profiles/r04_pmc_summary.txt"}}
f, w, n = tot("void k_ssor_color_split")
if n:
    ap = n / 40.0
    out.append("k_ssor_color_split: %d launches = %.1f applies x 40 colour sweeps; per apply fetch %.0f MB + write %.0f MB = %.0f MB vs algorithmic 7468 MB: %+.1f %%" % (n, ap, f / ap / 1e6, w / ap / 1e6, (f + w) / ap / 1e6, ((f + w) / ap / 7.468e9 - 1) * 100))
fe, we, ne = 0.0, 0.0, 0
for pre in ("void k_eis_backward_split", "void k_eis_forward_split", "void k_eis_update<1>"):
    a, b, c = tot(pre)
    fe += a; we += b
    if pre.startswith("void k_eis_update"):
        ne = c
if ne:
    out.append("Eisenstat iteration (k_eis_backward_split + k_eis_forward_split + k_eis_update<1>, %d iterations): fetch %.0f MB + write %.0f MB = %.0f MB vs algorithmic 9007 MB: %+.1f %%" % (ne, fe / ne / 1e6, we / ne / 1e6, (fe + we) / ne / 1e6, ((fe + we) / ne / 9.007e9 - 1) * 100))
f, w, n = tot("void k_assemble_c3d8<1>")
if n:
    per = 8.0
    t = (f + w) / n * per
    out.append("k_assemble_c3d8<1> (IC element, 8 colour launches per assembly, %d launches): per assembly fetch %.0f MB + write %.0f MB = %.0f MB vs algorithmic 30486 MB (64 read-modify-writes of 72 B per element): %+.1f %%" % (n, f / n * per / 1e6, w / n * per / 1e6, t / 1e6, (t / 30.486e9 - 1) * 100))
    traffic["k_assemble_c3d8"] = {"3307949": {"traffic_bytes": round(t), "fetch_bytes": round(f / n * per), "write_bytes": round(w / n * per), "source": "profiles/r04_pmc_summary.txt"}}
if ilu:
    out.append("")
    out.append("BiCGSTAB + ILU(0), 10.1M DOF (--pmc FETCH_SIZE pass of bench.py --method 2 --precond 10): FETCH_SIZE per launch")
    for k, v in sorted(ilu.items(), key=lambda kv: -sum(kv[1]))[:6]:
        m = sum(v) / len(v)
        out.append("  %-60s launches=%4d  %12.1f KiB -> %9.1f MB (x2 corrected)%s" % (k[:60], len(v), m, 2 * m * 1024 / 1e6, "   vs algorithmic 7468 MB: %+.1f %%" % ((2 * m * 1024 / 7.468e9 - 1) * 100) if "k_tri_dataflow" in k else ""))
open(os.path.join(D, "r04_pmc_summary.txt"), "w").write("\n".join(out) + "\n")
json.dump(traffic, open(os.path.join(D, "r04_traffic.json"), "w"), indent=1)
print("\n".join(out[-10:]))
