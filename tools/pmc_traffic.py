"""Summarise the separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over bench.py into
profiles/roundN/pmc_traffic.json (bench.py reads it for `roofline.traffic`): KB per launch, mean over the whole-layer launches
of the kernels named below.  Values are raw counter sums (FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950 under-reports
wide coalesced reads by 2x, MI355X_MICROARCH.md -- the FPS loads are 12-byte gathers, an uncalibrated width).

usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv
import json
import re
import sys

KERNELS = [  # key in the json, kernel-name regex, (batch, n, m) of the bench launch
    ("fps_pruned_kernel<32>", re.compile(r"fps_pruned_kernel<32, false, false, true"), (8, 16384, 4096)),
    ("sa_group_mlp_pm_kernel<128,128,256,256,2,32>", re.compile(r"sa_group_mlp_pm_kernel<128, 128, 256, 256, 2, 32"), (8, 1024, 512)),
    ("sa_group_mlp_pm_kernel<64,64,96,128,2,32>", re.compile(r"sa_group_mlp_pm_kernel<64, 64, 96, 128, 2, 32"), (8, 4096, 1024)),
    ("ball_query_dual_kernel", re.compile(r"ball_query_dual_kernel"), (8, 16384, 4096)),
    ("sa_group_mlp_f16_lds_kernel<128,256,256,1,32,4>", re.compile(r"sa_group_mlp_f16_lds_kernel<128, 256, 256, 1, 32, 4"), (8, 1024, 512)),
    ("sa_group_mlp_f16_kernel<64,96,2,32,lds-weights>", re.compile(r"sa_group_mlp_f16_kernel<64, 96, 2, 32, true"), (8, 4096, 1024)),
    ("ball_query_wave_multi_kernel<4,8,8>", re.compile(r"ball_query_wave_multi_kernel<4, 8, 8>"), (8, 16384, 4096)),
]


def per_launch(path, counter):
    acc = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            for key, rx, _ in KERNELS:
                if rx.search(row["Kernel_Name"]):
                    d = acc.setdefault(key, {})
                    rec = d.setdefault(row["Dispatch_Id"], [int(row["Grid_Size"]), 0.0])
                    rec[1] += float(row["Counter_Value"])
    out = {}
    for key, d in acc.items():
        big = max(g for g, _ in d.values())
        vals = [v for g, v in d.values() if g == big]
        out[key] = (sum(vals) / len(vals), len(vals))
    return out


def main():
    fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
    out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; tools/profile_round.sh) "
                       "over python3 bench.py --steps 3 --warmup 1 on MI355X; KB per launch, mean over the launches with the "
                       "largest grid; raw counter sums (see tools/pmc_traffic.py)"}
    for key, _, (b, n, m) in KERNELS:
        if key in fetch and key in write:
            out[key] = {"batch": b, "n": n, "m": m, "fetch_kb": round(fetch[key][0], 1), "write_kb": round(write[key][0], 1),
                        "launches": fetch[key][1]}
            print(f"{key:52s} fetch {fetch[key][0]:10.1f} KB  write {write[key][0]:10.1f} KB  ({fetch[key][1]} launches)")
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
