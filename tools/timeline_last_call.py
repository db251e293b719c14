"""Start / duration of every psk kernel of the LAST call in a rocprofv3 kernel trace (csv), relative to the call's first launch:
python tools/timeline_last_call.py <kernel_trace.csv> [first-kernel-substring]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "psk" in r["Kernel_Name"] and "probe" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "front"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# the last call begins at the last front-stage launch that follows a launch of another kind
begin = max(i for i in starts if i == 0 or first not in rows[i - 1]["Kernel_Name"] or True)
# (two halves: two front launches a call -- walk back to the first of the pair)
while begin > 0 and first in rows[begin - 1]["Kernel_Name"]:
    begin -= 1
t0 = int(rows[begin]["Start_Timestamp"])
for r in rows[begin:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("psk::", "")
    print("%-34.34s queue %-3s start %8.1f us  dur %7.1f us  end %8.1f" % (
        name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        (int(r["End_Timestamp"]) - t0) / 1e3))
