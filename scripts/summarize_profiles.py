#!/usr/bin/env python3
"""Copy the rocprofv3 outputs of scripts/profile_bench.sh (gpurun_out/prof) into profiles/ as the
round's committed summaries: kernel stats CSV + a PMC traffic summary with the gfx950 FETCH_SIZE correction."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prof = os.path.join(ROOT, "gpurun_out", "prof")
def newest(pat):
    fs = sorted(glob.glob(os.path.join(prof, pat)), key=os.path.getmtime)
    return fs[-1]
stats, fetch, write = newest("trace/*/*_kernel_stats.csv"), newest("pmc_fetch/*/*_counter_collection.csv"), newest("pmc_write/*/*_counter_collection.csv")
shutil.copy(stats, os.path.join(ROOT, "profiles", "%s_rocprofv3_kernel_stats_cg_ssor_10m.csv" % tag))
out = ["rocprofv3 PMC passes (separate runs, --pmc FETCH_SIZE / --pmc WRITE_SIZE; scripts/profile_bench.sh) of",
       "  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline   (10.125M DOF, CG + multicolour SSOR, MI355X)",
       "Counter unit: KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of",
       "wide coalesced streaming reads -> fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.", ""]
vals = {}
for name, f in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out.append(name + " per launch (mean over launches)")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
        m = sum(v) / len(v)
        b = m * 1024 * (2 if name == "FETCH_SIZE" else 1)
        vals[(name, k)] = (b, len(v))
        out.append("  %-46s launches=%4d  counter=%12.1f KiB  -> %8.1f MB%s" % (k[:46], len(v), m, b / 1e6, " (x2 corrected)" if name == "FETCH_SIZE" else ""))
    out.append("")
def tot(prefix):
    f = sum(b * n for (nm, k), (b, n) in vals.items() if nm == "FETCH_SIZE" and k.startswith(prefix))
    w = sum(b * n for (nm, k), (b, n) in vals.items() if nm == "WRITE_SIZE" and k.startswith(prefix))
    n = sum(n for (nm, k), (b, n) in vals.items() if nm == "FETCH_SIZE" and k.startswith(prefix))
    return f, w, n
f, w, n = tot("void k_spmv<0, 1")
spmv_traffic = (f + w) / n
out.append("k_spmv<0,1,...> per launch: fetch %.0f MB + write %.0f MB = %.0f MB HBM traffic vs algorithmic 7009 MB (SURVEY 8d): %+.1f %%" % (f / n / 1e6, w / n / 1e6, spmv_traffic / 1e6, (spmv_traffic / 7.009e9 - 1) * 100))
f2, w2, n2 = tot("void k_ssor_color")
ncol = 40
out.append("k_ssor_color: %d launches = %.1f applies x %d colour sweeps; per apply fetch %.0f MB + write %.0f MB = %.0f MB vs algorithmic 7468 MB: %+.1f %%" % (n2, n2 / ncol, ncol, f2 / (n2 / ncol) / 1e6, w2 / (n2 / ncol) / 1e6, (f2 + w2) / (n2 / ncol) / 1e6, ((f2 + w2) / (n2 / ncol) / 7.468e9 - 1) * 100))
open(os.path.join(ROOT, "profiles", "%s_pmc_summary.txt" % tag), "w").write("\n".join(out) + "\n")
json.dump({"_comment": "HBM traffic per launch from rocprofv3 PMC passes (profiles/%s_pmc_summary.txt): 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, gfx950 correction per MI355X_MICROARCH.md. Keyed by block rows of the workload." % tag,
           "k_spmv": {"3375000": {"traffic_bytes": round(spmv_traffic), "fetch_bytes": round(f / n), "write_bytes": round(w / n), "source": "profiles/%s_pmc_summary.txt" % tag}}},
          open(os.path.join(ROOT, "profiles", "%s_traffic.json" % tag), "w"), indent=1)
print("\n".join(out[-3:]))
