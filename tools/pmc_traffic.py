#!/usr/bin/env python3
"""Merge the FETCH_SIZE and WRITE_SIZE pass summaries of tools/profile_round.sh into the per-kernel traffic file bench.py reads.
   python tools/pmc_traffic.py gpurun_out/r02_pmc_FETCH_SIZE.json gpurun_out/r02_pmc_WRITE_SIZE.json profiles/r02_pmc_traffic.json "provenance text" [pairs per launch]
   The file records the sha of the kernel sources it was taken on (bench.kernel_source_sha) and the batch size: bench.py attaches a traffic figure only while both still match."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
f = json.load(open(sys.argv[1])); w = json.load(open(sys.argv[2]))
import bench
out = {"_provenance": sys.argv[4] if len(sys.argv) > 4 else "", "kernel_source_sha": bench.kernel_source_sha(), "pairs_per_launch": int(sys.argv[5]) if len(sys.argv) > 5 else 5000000, "kernels": {}}
for k in sorted(set(f) | set(w)):
    fe = f.get(k, {}).get("FETCH_SIZE"); wr = w.get(k, {}).get("WRITE_SIZE")
    out["kernels"][k] = {"launches": int((fe or wr)["launches"]),
                         "fetch_bytes_per_launch": int(fe["avg_per_launch"] * 1024) if fe else None,    # rocprofv3 reports KiB
                         "write_bytes_per_launch": int(wr["avg_per_launch"] * 1024) if wr else None}
json.dump(out, open(sys.argv[3], "w"), indent=0)
print("wrote", sys.argv[3], len(out["kernels"]), "kernels")
