#!/usr/bin/env python3
"""tools/pmc_summarise.py TAG KEY [--kernel SUBSTR]  (KEY = workload, or workload_pID for policy ID) — turn the raw rocprofv3 output of tools/profile_bench.sh
(gpurun_out/prof_TAG/) into the tracked evidence under profiles/:

    profiles/TAG_bench.json          the un-profiled bench.py line
    profiles/TAG_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary
    profiles/TAG_pmc.csv             per-counter averages over the timed launches of the dominant kernel, per launch and
                                     per env-step (FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them)
and refreshes profiles/pmc_summary.json[WORKLOAD], which bench.py reports as roofline.traffic and the `issue` block.
FETCH_SIZE is doubled for gfx950 coalesced streaming reads as MI355X_MICROARCH.md (HBM section) prescribes.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, workload = sys.argv[1], sys.argv[2]
    want = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--kernel" else None
    src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(REPO, "profiles")
    bench = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
    with open(os.path.join(dst, f"{tag}_bench.json"), "w") as f:
        json.dump(bench, f)
        f.write("\n")
    env_steps = bench["roofline"]["env_steps_per_launch"]
    stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    # dominant kernel = the one with the largest total time in the stats
    kname = want
    if stats and not kname:
        rows = list(csv.DictReader(open(stats[0])))
        rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total_Duration_Ns", 0)) or 0))
        kname = rows[0]["Name"]
    sums, counts = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(dict)
        for r in csv.DictReader(open(path)):
            if kname and kname.split("(")[0][:40] not in r["Kernel_Name"]:
                continue
            per_dispatch[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        for cname, d in per_dispatch.items():
            ids = sorted(d)
            # the bench runs: fill + warm-up launches, then the timed ones: keep the second half (steady state)
            keep = ids[len(ids) // 2:]
            sums[cname] += sum(d[i] for i in keep)
            counts[cname] += len(keep)
    avg = {c: sums[c] / counts[c] for c in sums if counts[c]}
    with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
        f.write(f"# rocprofv3 --pmc passes (one run per counter group, no tracing) of: python3 bench.py --no-cpu-baseline --steps 8 "
                f"[{bench['config']['workload'].split(':')[0]}]\n# kernel {kname}; {env_steps:.0f} env-steps per launch; averages over the "
                f"steady-state launches. FETCH_SIZE/WRITE_SIZE in KiB; SQ cycle counters in units of 4 clocks summed over waves\n")
        f.write("counter,per_launch,per_env_step\n")
        for c in sorted(avg):
            f.write(f"{c},{avg[c]:.6g},{avg[c] / env_steps:.6g}\n")
    per = {c: v / env_steps for c, v in avg.items()}
    summ = {"source": f"profiles/{tag}_pmc.csv (rocprofv3 --pmc, separate passes; kernel {kname})"}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        summ["fetch_kib_per_launch"] = avg["FETCH_SIZE"]
        summ["write_kib_per_launch"] = avg["WRITE_SIZE"]
        summ["hbm_bytes_per_env_step"] = (2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024 / env_steps
    for key, c in (("valu_per_env_step", "SQ_INSTS_VALU"), ("salu_per_env_step", "SQ_INSTS_SALU"),
                   ("lds_per_env_step", "SQ_INSTS_LDS"), ("smem_per_env_step", "SQ_INSTS_SMEM"),
                   ("vmem_per_env_step", "SQ_INSTS_VMEM_RD")):
        if c in per:
            summ[key] = per[c]
    if "SQ_WAVE_CYCLES" in avg:
        if "SQ_WAIT_ANY" in avg:
            summ["wait_any_frac"] = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
        summ["wave_cycles_per_env_step"] = per["SQ_WAVE_CYCLES"] * 4
    if "SQ_ACTIVE_INST_VALU" in avg:
        summ["valu_active_cycles_per_env_step"] = per["SQ_ACTIVE_INST_VALU"] * 4
    if "SQ_ACTIVE_INST_SCA" in avg:
        summ["salu_active_cycles_per_env_step"] = per["SQ_ACTIVE_INST_SCA"] * 4
    # ties the summary to the kernel sources it was measured on (bench.py marks it stale when they differ)
    sys.path.insert(0, REPO)
    import bench
    summ["kernel_source_sha"] = bench.kernel_source_hash()
    pj = os.path.join(dst, "pmc_summary.json")
    allp = json.load(open(pj)) if os.path.exists(pj) else {}
    allp[workload] = summ
    with open(pj, "w") as f:
        json.dump(allp, f, indent=2)
        f.write("\n")
    print(json.dumps(summ, indent=1))
    print({c: round(v, 3) for c, v in per.items()})


if __name__ == "__main__":
    main()
