"""Summarise gpurun_out/wait_<workload> (tools/collect_wait_profiles.sh) into profiles/<tag>_pmc_wait_<workload>.json: per-launch counter
means of the solve kernel and the derived attribution of the wait cycles and of the factor-record traffic.
Usage: python tools/summarise_wait_profiles.py r03 config3"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
wl = sys.argv[2] if len(sys.argv) > 2 else "config2"
src = os.path.join(ROOT, "gpurun_out", "wait_" + wl)
per, groups = {}, []
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
        groups.append(sorted(acc))
        for cname, v in acc.items():
            per.setdefault(cname, v / len(n[cname]))   # (SQ_WAVE_CYCLES rides in several passes: the first is kept)
g = lambda k: per.get(k, 0.0)
wc = g("SQ_WAVE_CYCLES")
B, N = {"config2": (256, 20), "config3": (4096, 20), "config5": (8192, 30)}[wl]
d = {}
if wc:
    d["wait_any_share"] = g("SQ_WAIT_ANY") / wc                 # wave parked: s_waitcnt (LDS / VMEM / SMEM returns) or s_barrier
    d["wait_inst_any_share"] = g("SQ_WAIT_INST_ANY") / wc       # wave ready, issue slot busy
    d["wait_inst_lds_share"] = g("SQ_WAIT_INST_LDS") / wc       # ... of which: LDS issue port busy
    d["active_inst_any_share"] = g("SQ_ACTIVE_INST_ANY") / wc
    d["active_inst_valu_share"] = g("SQ_ACTIVE_INST_VALU") / wc
    d["active_inst_lds_share"] = g("SQ_ACTIVE_INST_LDS") / wc
    d["active_inst_scalar_share"] = g("SQ_ACTIVE_INST_SCA") / wc
    d["active_inst_vmem_share"] = (g("SQ_ACTIVE_INST_VMEM") + g("SQ_ACTIVE_INST_FLAT")) / wc
    # SQ_INST_LEVEL_x accumulates the number of x instructions in flight every (quad-)cycle: level / count = mean latency, and
    # level / wave-cycles = mean number in flight per wave = an UPPER bound of the share of wave-cycles a wave can have spent
    # parked on x (it is parked on x only while at least one is in flight)
    if g("SQ_INSTS_LDS"):
        d["lds_mean_latency_quadcycles"] = g("SQ_INST_LEVEL_LDS") / g("SQ_INSTS_LDS")
    vm = g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")
    if vm:
        d["vmem_mean_latency_quadcycles"] = g("SQ_INST_LEVEL_VMEM") / vm
    if g("SQ_INSTS_SMEM"):
        d["smem_mean_latency_quadcycles"] = g("SQ_INST_LEVEL_SMEM") / g("SQ_INSTS_SMEM")
    d["lds_in_flight_per_wave"] = g("SQ_INST_LEVEL_LDS") / wc
    d["vmem_in_flight_per_wave"] = g("SQ_INST_LEVEL_VMEM") / wc
    d["smem_in_flight_per_wave"] = g("SQ_INST_LEVEL_SMEM") / wc
    d["lds_bank_conflict_share_of_lds_active"] = g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0)
if g("SQ_INSTS_MFMA") or g("SQ_INSTS_VALU_MFMA_F32"):
    # MFMA share: v_mfma_f32_16x16x4_f32 holds the matrix pipe of its SIMD for 32 cycles = 8 quad-cycles per instruction
    nm = g("SQ_INSTS_VALU_MFMA_F32") or g("SQ_INSTS_MFMA")
    d["mfma_instructions_per_launch"] = nm
    d["mfma_share_of_valu_instructions"] = nm / max(g("SQ_INSTS_VALU"), 1.0)
    d["mfma_pipe_quadcycles_over_wave_cycles"] = 8.0 * nm / max(wc, 1.0)
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
        d["mfma_busy_over_sq_busy_cycles"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")
hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
if hit + miss:
    d["l2_hit_rate"] = hit / (hit + miss)
    d["l2_requests_per_launch"] = g("TCC_REQ_sum")
    d["l2_read_requests"] = g("TCC_READ_sum"); d["l2_write_requests"] = g("TCC_WRITE_sum")
    d["l2_to_fabric_read_requests"] = g("TCC_EA0_RDREQ_sum"); d["l2_to_fabric_write_requests"] = g("TCC_EA0_WRREQ_sum")
    d["l1_to_l2_read_requests"] = g("TCP_TCC_READ_REQ_sum"); d["l1_to_l2_write_requests"] = g("TCP_TCC_WRITE_REQ_sum")
    d["l1_accesses"] = g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("TCP_TCC_READ_REQ_sum"):
        d["l1_to_l2_read_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
out = {"command": f"rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --workload {wl} --steps 5 --warmup 1 --no-cpu-baseline --secondary none "
                  f"(B={B}, N={N}); one counter group per pass (tools/collect_wait_profiles.sh)",
       "groups": groups, "per_launch": per, "derived": d,
       "note": "SQ_* cycle counters are quad-cycles summed over the waves of a launch; SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES. "
               "gfx950 lists no barrier-wait counter (rocprofv3 -L): the s_barrier share is what SQ_WAIT_ANY leaves after the memory returns, bounded as in DESIGN 6."}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_wait_{wl}.json"), "w"), indent=1)
print(json.dumps({"per_launch": per, "derived": d}, indent=1))
