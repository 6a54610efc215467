"""Summarise gpurun_out/prof (tools/collect_profiles.sh) into profiles/<tag>_kernel_stats.csv and
profiles/<tag>_pmc_summary.json.  Usage: python tools/summarise_profiles.py r01"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")

# (gpurun merges new output over old: several runs may lie side by side -- always the newest file)
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
assert stats, "no kernel_stats.csv under " + src
rows = list(csv.DictReader(open(stats[0])))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline  (config 2: B=256, N=20)\n")
    f.write(open(stats[0]).read())
solve = [r for r in rows if "cmpc_solve_kernel" in r["Name"]][0]
print("solve kernel:", solve["Name"][:60], "calls", solve["Calls"], "avg ns", solve["AverageNs"])

per = {}
for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for fn in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)[:1]:
        acc, n = {}, {}
        for r in csv.DictReader(open(fn)):
            if "cmpc_solve_kernel" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
        for cname, v in acc.items():
            per[cname] = v / len(n[cname])
B, N = 256, 20
nx, npar = 45 * N + 15, 50 * N + 27
alg = B * 4 * (nx + npar + nx + 8)   # P and X0 read, X and info written, once each
out = {
    "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline (config 2: B=256, N=20); one counter group per pass (tools/collect_profiles.sh)",
    "kernel": solve["Name"],
    "kernel_avg_ms_from_trace": float(solve["AverageNs"]) * 1e-6,
    "per_launch": per,
    "hbm_bytes_per_launch": (2.0 * per.get("FETCH_SIZE", 0.0) + per.get("WRITE_SIZE", 0.0)) * 1024.0,
    "algorithmic_bytes_per_launch": alg,
    "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated there on 16-byte-per-lane streams; "
            "our reads are 4 bytes per lane, so the doubled figure is an upper bound). SQ_* are summed over the 1024 waves of a launch (quad-cycles for *_CYCLES / WAIT counters).",
}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
