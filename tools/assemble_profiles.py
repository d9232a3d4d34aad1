#!/usr/bin/env python3
"""Assemble profiles/ from the output of tools/run_profiles.sh (gpurun_out/prof_round):
r01_kernel_stats.csv, r01_kernel_trace_summary.json, r01_pmc_summary.json, pmc_traffic.json."""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import pmc_summary  # noqa: E402
import profile_summary  # noqa: E402

GROUPS = {"c2": "32 channels (two 7.1 measurements) at circular length 540672", "c3": "13 channels at circular length 1179648",
          "c5": "8 channels at circular length 1572864"}


def main(out, tag="r01"):
    prof = os.path.join(ROOT, "profiles")
    line = json.load(open(os.path.join(out, "bench_trace.json")))
    cfg = line["config"]
    groups_per_step = cfg["measurements_per_step"] // cfg.get("measurements_per_launch_group", 1)
    n_timed, n_iso = line["steps"] * groups_per_step, max(4, min(40, groups_per_step))
    kt = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 "
                     "--no-cpu-baseline  (tools/run_profiles.sh)",
          "note": f"bench.py runs 2 warm-up steps, {line['steps']} timed steps (a step = {cfg['measurements_per_step']} resident "
                  f"measurements = {groups_per_step} launch groups of {cfg.get('channels_per_launch_group')} channels, 3 groups in "
                  f"flight) and {n_iso} strictly serial launch groups for the isolated roofline; the kernel_stats.csv averages "
                  "over ALL launches of each kernel, this file splits out the timed region and the serial tail",
          "kernels": profile_summary.main(os.path.join(out, "trace"), n_timed, n_iso)}
    json.dump(kt, open(os.path.join(prof, f"{tag}_kernel_trace_summary.json"), "w"), indent=1)
    for root, _, files in os.walk(os.path.join(out, "trace")):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                shutil.copy(os.path.join(root, f), os.path.join(prof, f"{tag}_kernel_stats.csv"))
    for root, _, files in os.walk(os.path.join(out, "trace_serial")):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                shutil.copy(os.path.join(root, f), os.path.join(prof, f"{tag}_kernel_stats_serial.csv"))
    for name in ("bench_trace.json", "bench_trace_serial.json"):
        src = os.path.join(out, name)
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(prof, f"{tag}_{name}"))
    traffic, pmc_all = {}, {}
    for w, what in GROUPS.items():
        d = os.path.join(out, f"pmc_{w}")
        if not os.path.isdir(d):
            continue
        pm = pmc_summary.main(d)
        pmc_all[w] = pm
        if not all(k in pm and "FETCH_SIZE" in pm[k] and "WRITE_SIZE" in pm[k] for k in ("rows_kernel", "cols_fwd", "cols_inv")):
            continue
        b = lambda k: (2 * pm[k]["FETCH_SIZE"] + pm[k]["WRITE_SIZE"]) * 1024      # noqa: E731
        traffic[w] = {"rows_kernel_bytes_per_launch": b("rows_kernel"), "cols_fwd_bytes_per_launch": b("cols_fwd"),
                      "cols_inv_bytes_per_launch": b("cols_inv"),
                      "rows_kernel_fetch_kb_raw": pm["rows_kernel"]["FETCH_SIZE"],
                      "rows_kernel_write_kb": pm["rows_kernel"]["WRITE_SIZE"],
                      "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + "
                              "WRITE_SIZE)*1024 per the gfx950 correction in MI355X_MICROARCH.md (FETCH_SIZE reports half of a "
                              "coalesced read stream; calibrated on the inverse column pass whose read volume is known); one launch = "
                              + what}
    json.dump(traffic, open(os.path.join(prof, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
                          "--workload W --lanes 1 --no-events --no-cpu-baseline  (tools/run_profiles.sh)",
               "unit": "KB per launch, raw counter values (FETCH_SIZE needs the x2 gfx950 correction)", "workloads": pmc_all},
              open(os.path.join(prof, f"{tag}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    print(json.dumps(kt["kernels"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r02")
