"""Summarise gpurun_out/prof (tools/collect_profiles.sh) into profiles/<tag>_kernel_stats.csv and
profiles/<tag>_pmc_summary.json.  Usage: python tools/summarise_profiles.py r01"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
wl = sys.argv[2] if len(sys.argv) > 2 else "config2"
src = os.path.join(ROOT, "gpurun_out", "prof_" + wl)
sfx = "" if wl == "config2" else "_" + wl
dst = os.path.join(ROOT, "profiles")

# (gpurun merges new output over old: several runs may lie side by side -- always the newest file)
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
assert stats, "no kernel_stats.csv under " + src
rows = list(csv.DictReader(open(stats[0])))
with open(os.path.join(dst, f"{tag}_kernel_stats{sfx}.csv"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --steps 20 --warmup 3 --no-cpu-baseline --secondary none\n")
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
B, N = {"config2": (256, 20), "config3": (4096, 20), "config5": (8192, 30)}[wl]
nx, npar = 45 * N + 15, 50 * N + 27
alg = B * 4 * (nx + npar + nx + 8)   # P and X0 read, X and info written, once each
out = {
    "command": f"rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --workload {wl} --steps 5 --warmup 1 --no-cpu-baseline --secondary none (B={B}, N={N}); one counter group per pass (tools/collect_profiles.sh)",
    "kernel": solve["Name"],
    "kernel_avg_ms_from_trace": float(solve["AverageNs"]) * 1e-6,
    "per_launch": per,
    "hbm_bytes_per_launch": (2.0 * per.get("FETCH_SIZE", 0.0) + per.get("WRITE_SIZE", 0.0)) * 1024.0,
    "algorithmic_bytes_per_launch": alg,
    "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated there on 16-byte-per-lane streams; "
            "our reads are 4 bytes per lane, so the doubled figure is an upper bound). SQ_* are summed over the 4 B waves of a launch (quad-cycles for *_CYCLES / WAIT counters).",
}
waves = 4 * B
pl = per
if pl.get("SQ_WAVE_CYCLES"):
    out["derived"] = {
        "wait_any_over_wave_cycles": pl.get("SQ_WAIT_ANY", 0.0) / pl["SQ_WAVE_CYCLES"],
        "valu_active_over_wave_cycles": pl.get("SQ_ACTIVE_INST_VALU", 0.0) / pl["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_cycles_per_lds_instruction": pl.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(pl.get("SQ_INSTS_LDS", 1.0), 1.0),
        "valu_instructions_per_launch": pl.get("SQ_INSTS_VALU", 0.0),
    }
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary{sfx}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
