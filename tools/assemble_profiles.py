"""Copies the judged summaries of one collection run (tools/collect_profiles.sh + tools/pmc_stalls.sh,
merged back under gpurun_out/) into profiles/<round>/<version>_* and refreshes profiles/traffic_latest.json.

usage: python tools/assemble_profiles.py gpurun_out/r01_v8 gpurun_out/r01_v8_stalls profiles/r01 v8
"""
import csv
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys

run, stalls, dst, ver = sys.argv[1:5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """gpurun merges into gpurun_out/: an earlier collection under the same tag may have left files behind."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def out(name):
    return os.path.join(dst, "%s_%s" % (ver, name))


shutil.copy(os.path.join(run, "bench.json"), out("bench.json"))
shutil.copy(os.path.join(run, "stream_floor.txt"), out("stream_floor.txt"))
shutil.copy(newest(os.path.join(run, "stats", "*", "*_kernel_stats.csv")), out("kernel_stats.csv"))

# per-launch trace of this library's kernels, and the mean over the timed region of that run
trace = newest(os.path.join(run, "stats", "*", "*_kernel_trace.csv"))
rows = list(csv.DictReader(open(trace)))
psk = [r for r in rows if "psk::" in r["Kernel_Name"]]
with open(out("kernel_trace_psk.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(psk)
fast = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in psk if "psk_fast_kernel<8, 1, false>" in r["Kernel_Name"]]
bench = json.loads(open(os.path.join(run, "bench.json")).read().strip().splitlines()[-1])
steps = bench["steps"]
with open(out("kernel_timed_region.txt"), "w") as f:
    f.write("psk_fast_kernel<8,1,false> under `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` (defaults: 30 warm-up + 50 timed steps)\n")
    f.write("launch durations (ms), in order: %s\n" % " ".join("%.3f" % x for x in fast))
    f.write("all %d launches: mean %.4f ms (this is AverageNs of %s)\n" % (len(fast), statistics.mean(fast), os.path.basename(out("kernel_stats.csv"))))
    f.write("warm-up launches (first %d): mean %.4f ms\n" % (len(fast) - steps, statistics.mean(fast[:-steps])))
    f.write("timed region (last %d): mean %.4f ms, min %.4f, max %.4f\n" % (steps, statistics.mean(fast[-steps:]), min(fast[-steps:]), max(fast[-steps:])))
    f.write("bench line of the unprofiled run on the same box (%s): launch_ms_avg %.4f (HIP events around upload + three kernels), ms_per_step %.4f\n"
            % (os.path.basename(out("bench.json")), bench["roofline"]["launch_ms_avg"], bench["ms_per_step"]))

summ = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + [os.path.join(run, d) for d in ("pmc_fetch", "pmc_write", "pmc_sq")],
                      capture_output=True, text=True, check=True).stdout
open(out("pmc_summary.txt"), "w").write(summ)
shutil.copy(os.path.join(stalls, "summary.txt"), out("pmc_stalls.txt"))


def median_of(counter, pass_dir):
    f = newest(os.path.join(run, pass_dir, "*", "*_counter_collection.csv"))
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
         if r["Counter_Name"] == counter and "psk_fast_kernel<8, 1, false>" in r["Kernel_Name"]]
    # one row per (dispatch, xcd/se instance) or per dispatch, depending on the rocprofv3 build: sum per dispatch
    return v


def per_dispatch(counter, pass_dir):
    f = newest(os.path.join(run, pass_dir, "*", "*_counter_collection.csv"))
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "psk_fast_kernel<8, 1, false>" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return statistics.median(acc.values())


fetch_kb, write_kb = per_dispatch("FETCH_SIZE", "pmc_fetch"), per_dispatch("WRITE_SIZE", "pmc_write")
json.dump({
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "fetch_size_kb_median": fetch_kb,
    "write_size_kb_median": write_kb,
    "note": "psk_fast_kernel<8,1,false>, steady-state launches of bench.py; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
            "half the bytes of wide coalesced reads; the pure-read probe kernel of the same run reads 8.59 GB and reports 4.29 GB), "
            "WRITE_SIZE as is; separate --pmc passes (%s)" % out("pmc_summary.txt"),
    "kernel": "psk_fast_kernel<8,1,false>",
}, open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic_latest.json"), "w"), indent=1)
print(open(out("kernel_timed_region.txt")).read())
print(open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic_latest.json")).read())
