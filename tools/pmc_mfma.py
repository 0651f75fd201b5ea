"""Summarise a rocprofv3 --pmc pass over bench.py into profiles/roundN/pmc_mfma.json: per grouped-MLP kernel the mean
launch duration, SQ_VALU_MFMA_BUSY_CYCLES and the MFMA-busy fraction = busy cycles / (duration x 1024 SIMDs x 2.4 GHz)
(256 CUs x 4 SIMDs; 2.4 GHz = the peak clock the dense MFMA peak is quoted at, so the fraction is against that peak).
With GRBM_GUI_ACTIVE in the pass also the clock the launch really ran at and the busy fraction of the elapsed cycles.

usage: python tools/pmc_mfma.py <counter_collection.csv> <precision: fp32|fp16x2|fp16> <out.json> [more csv:precision ...]
Entries are keyed "<precision>:<c1>,<c2>,ns<nsample>" (bench.py reads them for `roofline_mlp`)."""
import csv
import json
import re
import sys

SIMDS, CLOCK_HZ, XCDS = 1024, 2.4e9, 8
PATTERNS = [  # kernel-name regex -> (c1, c2, nsample) groups; trailing template arguments (flags) are free
    (re.compile(r"sa_group_mlp_pm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)[^>]*>"), lambda m: (m[2], m[3], m[6])),
    (re.compile(r"sa_group_mlp_f16_lds_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)[^>]*>"), lambda m: (m[1], m[2], m[5])),
    (re.compile(r"sa_group_mlp_f16_kernel<(\d+), (\d+), (\d+), (\d+)[^>]*>"), lambda m: (m[1], m[2], m[4])),
    (re.compile(r"sa_group_mlp_kernel<(\d+), (\d+), (\d+), (\d+)[^>]*>"), lambda m: (m[1], m[2], m[4])),
]


def summarise(path, precision, out):
    per = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            for rx, pick in PATTERNS:
                m = rx.search(name)
                if m:
                    c1, c2, ns = pick(m)
                    key = f"{precision}:{c1},{c2},ns{ns}"
                    rec = per.setdefault(key, {"kernel": name.split("(")[0].replace("void ", ""), "disp": {}})
                    d = rec["disp"].setdefault(row["Dispatch_Id"], {"dur": int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                                                    "grid": int(row["Grid_Size"])})
                    d[row["Counter_Name"]] = float(row["Counter_Value"])
                    break
    for key, rec in per.items():
        disp = list(rec["disp"].values())
        big = max(d["grid"] for d in disp)
        disp = [d for d in disp if d["grid"] == big and "SQ_VALU_MFMA_BUSY_CYCLES" in d]   # the whole-layer launches
        if not disp:
            continue
        dur = sum(d["dur"] for d in disp) / len(disp)
        busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in disp) / len(disp)
        entry = {"kernel": rec["kernel"], "launches": len(disp), "grid": big, "duration_us": dur / 1e3,
                 "mfma_busy_cycles": busy, "mfma_busy_frac": busy / (dur * 1e-9 * SIMDS * CLOCK_HZ)}
        gui = [d["GRBM_GUI_ACTIVE"] for d in disp if "GRBM_GUI_ACTIVE" in d]
        if gui:   # cycles the GPU was busy during the launch = the clock it really ran at (counters lower it)
            cyc = sum(gui) / len(gui) / XCDS   # the counter is summed over the 8 XCDs
            entry["grbm_gui_active_per_xcd"] = cyc
            entry["clock_ghz"] = cyc / dur
            entry["mfma_busy_frac_of_elapsed_cycles"] = busy / (cyc * SIMDS)
        for extra in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
            vals = [d[extra] for d in disp if extra in d]
            if vals:
                entry[extra.lower()] = sum(vals) / len(vals)
        out[key] = entry


def main():
    if len(sys.argv) < 4:
        raise SystemExit(__doc__)
    out_path = sys.argv[3]
    out = {"_comment": "rocprofv3 --pmc (own pass, --kernel-trace only) over bench.py on MI355X; profiled durations run "
                       "slower than un-profiled ones (lower clock under counters); mfma_busy_frac = "
                       "SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs x 2.4 GHz); clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / "
                       "duration (it includes the dispatch ramp, so short launches read high), mfma_busy_frac_of_elapsed_cycles "
                       "= busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"}
    summarise(sys.argv[1], sys.argv[2], out)
    for extra in sys.argv[4:]:
        path, prec = extra.rsplit(":", 1)
        summarise(path, prec, out)
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in out.items():
        if k != "_comment":
            print(f"{k:28s} {v['duration_us']:8.1f} us  mfma busy {100 * v['mfma_busy_frac']:5.1f} % of the 2.4 GHz peak"
                  + (f", {100 * v['mfma_busy_frac_of_elapsed_cycles']:5.1f} % of the elapsed cycles at {v['clock_ghz']:.2f} GHz"
                     if "clock_ghz" in v else "") + f"  {v['kernel']}")


if __name__ == "__main__":
    main()
