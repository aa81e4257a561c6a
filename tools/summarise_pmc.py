#!/usr/bin/env python3
"""Summarises two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in separate runs with
--kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/<tag>_pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python tools/summarise_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write r01

Units / corrections (guide, section HBM): counter values are KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide coalesced (16 B/lane) streaming read -> doubled here (checked on k_stream<...,false>, a pure read of
the 4.8 GB tensor: 2400.1 MB raw); WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: dict(launches=n, kib_total=v) for k, (n, v) in agg.items()}


def main():
    fetch, write, tag = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    out = {"note": "HBM bytes per LAUNCH; FETCH_SIZE doubled (gfx950 half-count of wide coalesced reads), KiB -> bytes",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fl = fetch.get(k, dict(launches=0, kib_total=0.0))
        wl = write.get(k, dict(launches=0, kib_total=0.0))
        n = max(fl["launches"], wl["launches"], 1)
        rd = 2.0 * fl["kib_total"] * 1024 / n
        wr = wl["kib_total"] * 1024 / n
        out["kernels"][k] = dict(launches=n, read_bytes=rd, write_bytes=wr, hbm_bytes=rd + wr,
                                 fetch_size_raw_kib=fl["kib_total"] / n, write_size_raw_kib=wl["kib_total"] / n)
    path = "profiles/%s_pmc_traffic.json" % tag
    json.dump(out, open(path, "w"), indent=1)
    for k in [q for q in out["kernels"] if q.startswith("k_project_l2d<")] + ["k_project_l2w<4, 1, 2, 3, 2, 1, 0>", "k_project_l2w<4, 1, 2, 2, 2, 1, 0>", "k_project_l2s<4, 2, 2, 1, 1>", "k_project_l2s<4, 2, 2, 1>", "k_project_l2<4, 2>", "k_project_lds", "k_project_mfma<8, 16>", "k_stream<256, 4, true>", "k_stream<256, 4, false>"]:
        if k in out["kernels"]:
            print(k, "%.1f MB / launch" % (out["kernels"][k]["hbm_bytes"] / 1e6))
    print("wrote", path)


if __name__ == "__main__":
    main()
