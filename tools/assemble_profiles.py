#!/usr/bin/env python3
"""Assemble profiles/ from the output of tools/run_profiles.sh (gpurun_out/prof_round): rNN_kernel_stats.csv (+ _serial),
rNN_pmc_summary.json, rNN_pmc_traffic.json and the bench lines of the round."""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import pmc_summary  # noqa: E402

CALLS = {"c2": "one chain call = 32 channels (two 7.1 measurements) at circular length 540672 (one-channel-per-transform plan), FIR stage 32640 (*) 9600",
         "c3": "one chain call = 26 channels at circular length 1179648 (pair-mode plan, 288 rows), FIR stage 65280 (*) 19200",
         "c5": "K1 alone, launch groups of 8 channels at circular length 1572864 (pair-mode plan, 384 rows)"}


def main(out, tag="r04"):
    prof = os.path.join(ROOT, "profiles")
    for name, dst in (("slice_rate_serial.txt", f"{tag}_slice_rate_serial.txt"), ("slice_rate.txt", f"{tag}_slice_rate.txt"),
                      ("eq_fir.txt", f"{tag}_eq_fir.txt"), ("slice_stages.txt", f"{tag}_slice_stages.txt"),
                      ("column_error.txt", f"{tag}_column_error.txt"), ("slice_rate_decay_serial.txt", f"{tag}_slice_rate_decay_serial.txt"),
                      ("slice_e2e.txt", f"{tag}_slice_e2e.txt"), ("link_probe.txt", f"{tag}_link_probe.txt")):
        src = os.path.join(out, name)
        if os.path.exists(src) and os.path.getsize(src):
            with open(src) as fh:
                text = "".join(ln for ln in fh if "simple_timer.cpp" not in ln and "UserWarning" not in ln and "warnings.warn" not in ln)
            open(os.path.join(prof, dst), "w").write(text)
    for sub, name in (("trace", f"{tag}_kernel_stats.csv"), ("trace_serial", f"{tag}_kernel_stats_serial.csv"),
                      ("trace_slice_serial", f"{tag}_slice_kernel_stats_serial.csv"), ("trace_slice", f"{tag}_slice_kernel_stats.csv"),
                      ("trace_slice_decay", f"{tag}_slice_decay_kernel_stats_serial.csv"), ("trace_eq", f"{tag}_eq_fir_kernel_stats.csv")):
        for root, _, files in os.walk(os.path.join(out, sub)):
            for f in files:
                if f.endswith("kernel_stats.csv"):
                    shutil.copy(os.path.join(root, f), os.path.join(prof, name))
    for name in ("bench_trace.json", "bench_trace_serial.json", "bench_c2.json", "bench_c2_steps20.json", "bench_c3.json", "bench_c5.json"):
        src = os.path.join(out, name)
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(prof, f"{tag}_{name}"))
    traffic, pmc_all = {}, {}
    for w, what in CALLS.items():
        d = os.path.join(out, f"pmc_{w}")
        if not os.path.isdir(d):
            continue
        pm = pmc_summary.main(d)
        if "rows_kernel" not in pm and "rows_single" in pm:
            pm["rows_kernel"] = pm["rows_single"]              # pair-mode plans: rows_single_kernel is the row pass
        pmc_all[w] = pm
        t = {}
        for k in ("cols_fwd", "rows_kernel", "cols_inv", "peak_search", "fir_block"):
            if k in pm and "FETCH_SIZE" in pm[k] and "WRITE_SIZE" in pm[k]:
                t[k + "_bytes_per_launch"] = (2 * pm[k]["FETCH_SIZE"] + pm[k]["WRITE_SIZE"]) * 1024
        if "rows_kernel" in pm:
            t["rows_kernel_fetch_kb_raw"] = pm["rows_kernel"].get("FETCH_SIZE")
            t["rows_kernel_write_kb"] = pm["rows_kernel"].get("WRITE_SIZE")
        t["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --pmc-child (strictly serial "
                     "chain calls); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction in MI355X_MICROARCH.md "
                     "(FETCH_SIZE reports half of a coalesced read stream); " + what)
        traffic[w] = t
    json.dump(traffic, open(os.path.join(prof, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
                          "--workload W --pmc-child  (tools/run_profiles.sh)",
               "unit": "KB per launch, raw counter values (FETCH_SIZE needs the x2 gfx950 correction)", "workloads": pmc_all},
              open(os.path.join(prof, f"{tag}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r04")
