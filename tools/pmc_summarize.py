#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc CSVs to per-kernel average FETCH_SIZE / WRITE_SIZE and apply the calibration."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
GIB = float(1 << 30)


def per_kernel(prefix, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, "**", "%s_%s_counter_collection.csv" % (prefix, counter)), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


res = {"note": "counter values are averages per dispatch in the counter's native unit (KiB per rocprofv3 docs); "
               "scale = true_bytes / (counter*1024) measured on calibration kernels moving exactly 1 GiB"}
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    pk = per_kernel("calib", c)
    cal[c] = {k: v[0] for k, v in pk.items()}
res["calibration_raw"] = cal
scale = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in cal[c].items():
        if ("copy<double>" in k and c == "FETCH_SIZE") or ("write<double>" in k and c == "WRITE_SIZE"):
            scale[c + "_f64"] = GIB / (v * 1024.0) if v else None
        if ("copy<float>" in k and c == "FETCH_SIZE") or ("write<float>" in k and c == "WRITE_SIZE"):
            scale[c + "_f32"] = GIB / (v * 1024.0) if v else None
res["scale"] = scale
for prefix, key, sfx in (("n4096", "n4096", "_f64"), ("n8192", "n8192", "_f64"), ("n262144", "n262144", "_f64"), ("n262144dyn", "n262144", "_f64"),
                         ("n32768f32", "n32768_f32", "_f32"), ("n262144f32", "n262144_f32", "_f32"), ("n262144dynf32", "n262144_f32", "_f32")):
    r = res.get(key, {})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, (avg, cnt) in per_kernel(prefix, c).items():
            e = r.setdefault(k, {})
            e[c + "_raw_KiB"] = avg
            e["dispatches_" + c] = cnt
            s = scale.get(c + sfx)
            e[c + "_bytes_corrected"] = avg * 1024.0 * s if s else None
    for k, e in r.items():
        if e.get("FETCH_SIZE_bytes_corrected") is not None and e.get("WRITE_SIZE_bytes_corrected") is not None:
            e["hbm_bytes_per_launch"] = e["FETCH_SIZE_bytes_corrected"] + e["WRITE_SIZE_bytes_corrected"]
    res[key] = r
print(json.dumps(res, indent=1))
