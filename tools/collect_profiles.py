#!/usr/bin/env python3
"""Turn the rocprofv3 output of a round (gpurun_out/<prefix>_{kt,fetch,write,sq,grbm}) into the tracked
summaries under profiles/.  usage: tools/collect_profiles.py prof3 r01"""
import collections, csv, glob, json, os, shutil, sys
pre, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out"); pr = os.path.join(root, "profiles")
shutil.copy(glob.glob(f"{go}/{pre}_kt/*/*_kernel_stats.csv")[0], f"{pr}/{rnd}_bench_kernel_stats.csv")
def counters(tag):
    f = glob.glob(f"{go}/{pre}_{tag}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: dict(dispatches=len(v), mean=sum(v) / len(v), max=max(v)) for c, v in cs.items()} for k, cs in agg.items()}
fetch, write = counters("fetch"), counters("write")
json.dump({"unit": "KB per dispatch", "FETCH_SIZE": {k: v["FETCH_SIZE"] for k, v in fetch.items()},
           "WRITE_SIZE": {k: v["WRITE_SIZE"] for k, v in write.items()}}, open(f"{pr}/{rnd}_bench_pmc_fetch_write.json", "w"), indent=1)
main = [k for k in fetch if "k_fused_fast<4, true" in k][0]
burn = [k for k in fetch if "k_fused_fast<4, false" in k][0]
var = [k for k in fetch if "k_variance" in k][0]
t = lambda k: (2 * fetch[k]["FETCH_SIZE"]["mean"] + write[k]["WRITE_SIZE"]["mean"]) * 1024
json.dump({"round": rnd, "source": f"profiles/{rnd}_bench_pmc_fetch_write.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0)",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; the factor 2 on FETCH_SIZE is the gfx950 correction (MI355X_MICROARCH.md, HBM), calibrated in the same run: "
                         "k_variance reads 4 MiB and reports %.1f KB, writes 4 MiB and reports %.1f KB" % (fetch[var]["FETCH_SIZE"]["mean"], write[var]["WRITE_SIZE"]["mean"]),
           "k_fused_steps_main_bytes_per_launch": t(main), "k_fused_steps_burn_bytes_per_launch": t(burn), "kernel_main": main},
          open(f"{pr}/pmc_traffic.json", "w"), indent=1)
sq = counters("sq")[main]; grbm = counters("grbm")[main]
res = {c: v["mean"] for c, v in sq.items()}
res["GRBM_GUI_ACTIVE"] = grbm["GRBM_GUI_ACTIVE"]["mean"]
res["valu_busy_fraction"] = 4 * res["SQ_ACTIVE_INST_VALU"] / 1024 / (res["GRBM_GUI_ACTIVE"] / 8)
res["valu_instructions_per_wave_step"] = res["SQ_INSTS_VALU"] / (4096 * 250.0)
res["_note"] = (main + ": mean per launch (250 steps x 65536 chains, 4096 waves); SQ_* busy/wait counters are in quad-cycles; "
                "GRBM_GUI_ACTIVE is summed over the 8 XCDs. valu_busy_fraction = 4*SQ_ACTIVE_INST_VALU/1024 SIMDs over GRBM_GUI_ACTIVE/8 kernel cycles.")
json.dump(res, open(f"{pr}/{rnd}_fused_kernel_sq_counters.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in ("valu_busy_fraction", "valu_instructions_per_wave_step")}), t(main))
