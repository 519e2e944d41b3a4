#!/usr/bin/env python3
"""rocprofv3 --pmc passes of the benched gather kernel (tools/final_refresh.sh: one counter group per pass, each under
<dir>/pmc_<group>/) -> the per-launch means bench.py reads for `roofline.traffic` (profiles/rNN/pmc_msda_fwd_B4_encoder.json).
    python3 tools/pmc_to_json.py <dir with pmc_*/> <commit> > pmc_msda_fwd_B4_encoder.json"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    root, commit = sys.argv[1], sys.argv[2]
    config = sys.argv[3] if len(sys.argv) > 3 else "r50"
    # the kernel bench.py's roofline entry times: the resident-levels kernel at the R50 shape, the query-run kernel at FocalNet's
    pattern = "msda_fwd_res_kernel" if config == "r50" else "msda_fwd_qrun_kernel"
    acc = defaultdict(list)
    for path in glob.glob(root + "/pmc_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if pattern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
    f, w = mean["FETCH_SIZE"], mean["WRITE_SIZE"]
    json.dump({
        "kernel": ("msda_fwd_res_kernel<L=4, FUSED=false> (csrc/msda_res.hip), head-major value [B,H,S,D] (the benched roofline kernel), "
                   "encoder shape B=4 S=Nq=22323 L=4") if config == "r50" else
                  ("msda_fwd_qrun_kernel<bf16, L=5, FUSED=false>, head-major value [B,H,S,D] (the roofline kernel of --config focalnet), "
                   "encoder shape B=2 S=Nq=204098 L=5"),
        "commit": commit,
        "command": "rocprofv3 --pmc <one group per pass> -- python3 tools/profile_win.py bhsd 6 auto" + ("" if config == "r50" else " focalnet"),
        "per_launch_mean": {"bf16": mean},
        "launches_per_pass": {k: len(v) for k, v in sorted(acc.items())},
        "notes": "FETCH_SIZE / WRITE_SIZE in KiB, each in its own pass; TCP_TCC_READ_REQ in requests; *_sum and SQ_* summed over "
                 "the chip. HBM traffic per launch = 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md, HBM section) + "
                 "WRITE_SIZE = %.1f MB against %s MB algorithmic (raw sum %.1f MB)" % ((2 * f + w) * 1024 / 1e6, "228.6" if config == "r50" else "1201.7",
                                                                                      (f + w) * 1024 / 1e6),
    }, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
