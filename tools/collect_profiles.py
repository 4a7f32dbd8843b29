#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/make_profiles.sh and copy it into profiles/<tag>/.

    python3 tools/collect_profiles.py gpurun_out/profiles_r01 summarize   # on the GPU box (writes summary.json there)
    python3 tools/collect_profiles.py gpurun_out/profiles_r01 collect r01 # here: copy into profiles/r01 + pmc_traffic.json

HBM traffic per launch (the `roofline.traffic` field of bench.py), following
MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots":
  * FETCH_SIZE and WRITE_SIZE are reported in KiB and come from separate passes;
  * on gfx950 FETCH_SIZE reads exactly 1/2 of a wide (16 B/lane) coalesced streaming read -> x2 for the
    phase-1 depth stream, whose reported size is isolated with the phase-1-only entry (tsdf_aabb_hip);
  * the remaining reads are the 4 B/lane staging copy (uncalibrated width): they are priced at their
    known byte count (sum over frames of the valid-pixel rectangle, computed from the same seeded
    frames), and the reported number is kept next to it;
  * WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import csv
import glob
import importlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "tsdf_fused_kernel<32, 0, false, false"   # (prefix: the group count follows as a fifth template argument)


def counter_medians(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = {"median": float(np.median(v)), "n": len(v)}
    return out


def kernel_stats(d):
    for f in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Name"]:
                return {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                        "max_ns": float(r["MaxNs"]), "stddev_ns": float(r["StdDev"]), "file": os.path.basename(f)}
    return None


def staging_bytes():
    sys.path.insert(0, ROOT)
    synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
    tot = 0
    for i in range(1024):
        h, d = synth.synth_frame(i, "full")
        img = np.abs(d.reshape(h[5] - h[3], h[4] - h[2])) >= 1
        ys, xs = np.nonzero(img)
        tot += int((ys.max() - ys.min() + 1) * (xs.max() - xs.min() + 1)) * 4
    return tot


def summarize(out):
    s = {"kernel": KERNEL}
    s["kernel_trace"] = kernel_stats(os.path.join(out, "trace"))
    fetch = counter_medians(os.path.join(out, "pmc_fetch")).get("FETCH_SIZE")
    write = counter_medians(os.path.join(out, "pmc_write")).get("WRITE_SIZE")
    fetch_aabb = counter_medians(os.path.join(out, "pmc_fetch_aabb")).get("FETCH_SIZE")
    s["pmc"] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "FETCH_SIZE_KiB_phase1_only": fetch_aabb}
    s["sq"] = counter_medians(os.path.join(out, "pmc_sq"))
    if fetch and write and fetch_aabb:
        wide = fetch_aabb["median"] * 1024 * 2          # 16 B/lane stream: reported at 1/2
        narrow_reported = (fetch["median"] - fetch_aabb["median"]) * 1024
        stage = staging_bytes()
        wr = write["median"] * 1024
        s["traffic"] = {
            "depth_stream_bytes": wide, "staging_bytes_known": stage, "staging_bytes_reported": narrow_reported,
            "write_bytes": wr, "hbm_bytes_per_launch": wide + stage + wr,
        }
    try:
        s["bench_unprofiled"] = json.loads(open(os.path.join(out, "bench_unprofiled.json")).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        s["bench_unprofiled"] = str(e)
    json.dump(s, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(json.dumps({k: s[k] for k in ("kernel_trace", "pmc")}, indent=1))
    if "traffic" in s:
        print(json.dumps(s["traffic"], indent=1))


def collect(out, tag):
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    # gpurun merges results into gpurun_out/ without deleting what earlier calls left there: take the trace
    # files of the run the summary was computed from (its pid prefix), not whatever the glob finds last
    summ = json.load(open(os.path.join(out, "summary.json")))
    prefix = summ.get("kernel_trace", {}).get("file", "").split("_", 1)[0]
    for sub, pat in (("trace", "*_kernel_stats.csv"), ("trace", "*_domain_stats.csv")):
        for f in sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True), key=os.path.getmtime):
            if prefix and not os.path.basename(f).startswith(prefix + "_"):
                continue
            shutil.copy(f, os.path.join(dst, "bench_" + os.path.basename(f).split("_", 1)[1]))
    for sub in ("pmc_fetch", "pmc_write", "pmc_fetch_aabb", "pmc_sq"):
        files = sorted(glob.glob(os.path.join(out, sub, "**", "*_counter_collection.csv"), recursive=True),
                       key=os.path.getmtime)
        for f in files[-1:]:  # the newest run only (see above)
            rows = [r for r in csv.DictReader(open(f)) if "tsdf" in r["Kernel_Name"]]
            keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                    "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
            with open(os.path.join(dst, sub + "_counters.csv"), "w", newline="") as g:
                w = csv.DictWriter(g, fieldnames=keep)
                w.writeheader()
                for r in rows:
                    w.writerow({k: r[k] for k in keep})
    for sub in ("aug64/trace", "crop_trace", "r64_trace"):
        for f in sorted(glob.glob(os.path.join(out, sub, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            shutil.copy(f, os.path.join(dst, sub.split("/")[0] + "_kernel_stats.csv"))
    for sub in ("aug64/sq_a", "aug64/sq_b", "aug64/sq_c", "crop_fetch", "crop_write"):
        files = sorted(glob.glob(os.path.join(out, sub, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
        for f in files[-1:]:
            rows = [r for r in csv.DictReader(open(f)) if "tsdf" in r["Kernel_Name"]]
            keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                    "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
            with open(os.path.join(dst, sub.replace("/", "_") + "_counters.csv"), "w", newline="") as g:
                w = csv.DictWriter(g, fieldnames=keep)
                w.writeheader()
                for r in rows:
                    w.writerow({k: r[k] for k in keep})
    for f in ("ab_fill_policy.log",):
        if os.path.exists(os.path.join(out, f)):
            shutil.copy(os.path.join(out, f), os.path.join(dst, f))
    for f in ("summary.json", "bench_unprofiled.json"):
        if os.path.exists(os.path.join(out, f)):
            shutil.copy(os.path.join(out, f), os.path.join(dst, f))
    s = json.load(open(os.path.join(out, "summary.json")))
    if "traffic" in s:
        t = dict(s["traffic"])
        t["source"] = f"profiles/{tag}/summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        json.dump(t, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print("copied into", dst)


def summarize2(out):
    """Second half: the augmented-kernel and crop passes, merged into summary.json."""
    s = json.load(open(os.path.join(out, "summary.json")))
    try:
        s["aug64"] = json.load(open(os.path.join(out, "aug64", "summary.json")))
    except Exception as e:  # noqa: BLE001
        s["aug64"] = str(e)
    # plain 64^3: whole-run average and steady state (dispatches 41..: see tools/exp_pmc.py)
    r64 = {}
    for f in glob.glob(os.path.join(out, "r64_trace", "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "tsdf_fused_kernel<64" in r["Name"]:
                r64 = {"name": r["Name"][:80], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                       "min_us": float(r["MinNs"]) / 1e3}
    for f in glob.glob(os.path.join(out, "r64_trace", "**", "*_kernel_trace.csv"), recursive=True):
        d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))
                   if "tsdf_fused_kernel<64" in r["Kernel_Name"])
        d = [(e - b) / 1e3 for b, e in d]
        if len(d) > 60 and r64:
            r64.update({"steady_calls": len(d) - 40, "steady_avg_us": float(np.mean(d[40:])), "first_10_avg_us": float(np.mean(d[:10]))})
    s["r64"] = r64
    crop = {}
    for f in glob.glob(os.path.join(out, "crop_trace", "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "tsdf_fused" in r["Name"]:
                crop["kernel_trace"] = {"name": r["Name"][:80], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                                        "min_ns": float(r["MinNs"])}
    for sub, cname in (("crop_fetch", "FETCH_SIZE"), ("crop_write", "WRITE_SIZE")):
        for f in glob.glob(os.path.join(out, sub, "**", "*_counter_collection.csv"), recursive=True):
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                 if "tsdf_fused" in r["Kernel_Name"] and r["Counter_Name"] == cname]
            if v:
                crop[cname + "_KiB_median"] = float(np.median(v))
    s["crops_1024"] = crop
    ab = os.path.join(out, "ab_fill_policy.log")
    if os.path.exists(ab):
        s["ab_fill_policy"] = open(ab).read().strip().splitlines()
    json.dump(s, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(json.dumps({k: s[k] for k in ("crops_1024",)}, indent=1))


def results_table(tag):
    """DESIGN.md's rocprofv3 column, printed from the TRACKED summary (profiles/<tag>/summary.json) so that the document
    and the profile cannot disagree:  python3 tools/collect_profiles.py - table r04"""
    s = json.load(open(os.path.join(ROOT, "profiles", tag, "summary.json")))
    rows = []
    kt = s.get("kernel_trace") or {}
    if kt:
        rows.append(("1024 full frames -> 32^3 (bench.py under the profiler)", kt["calls"], kt["average_ns"] / 1e3,
                     kt["min_ns"] / 1e3, None))
    for key, label in (("r64", "1024 full frames -> 64^3, plain"),):
        r = s.get(key) or {}
        if r:
            rows.append((label, r["calls"], r["avg_us"], r["min_us"], r.get("steady_avg_us")))
    for kn, r in ((s.get("aug64") or {}).get("kernel_trace") or {}).items():
        if "<64, 0, true" in kn:
            rows.append(("1024 full frames -> 64^3, augmented", r["calls"], r["avg_us"], r["min_us"], r.get("steady_avg_us")))
    c = (s.get("crops_1024") or {}).get("kernel_trace")
    if c:
        rows.append(("1024 crops -> 32^3", c["calls"], c["average_ns"] / 1e3, c["min_ns"] / 1e3, None))
    print("| workload | calls | rocprofv3 average (us) | min (us) | average over dispatches 41.. (us) |")
    print("|---|---|---|---|---|")
    for label, calls, avg, mn, steady in rows:
        print(f"| {label} | {calls} | {avg:.1f} | {mn:.1f} | {'%.1f' % steady if steady else '—'} |")


if __name__ == "__main__":
    if sys.argv[2] == "table":
        results_table(sys.argv[3])
    elif sys.argv[2] == "summarize":
        summarize(sys.argv[1])
    elif sys.argv[2] == "summarize2":
        summarize2(sys.argv[1])
    else:
        collect(sys.argv[1], sys.argv[3])
