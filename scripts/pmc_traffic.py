#!/usr/bin/env python3
"""gpurun_out/<tag>_{FETCH,WRITE}_SIZE.json (scripts/pmc_traffic.sh: per-kernel counter sums of the last full step) ->
profiles/r03_traffic.json (what bench.py reports as roofline.traffic / roofline_decoder.traffic) + a readable table.
HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes; scattered 4-byte reads are single 64-byte requests, so for the sampler / key kernels the doubled
figure is an upper bound - the raw one is listed beside it).
Usage: python scripts/pmc_traffic.py gpurun_out/r3pmc profiles/r03"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
KIB = 1024.0
DEC = ("dc_", "rocprim", "neg_sample")   # decoder kernels of a step: code-form fwd / bwd, pair list, its incidence sort
out = {"nodes": 1_000_000, "edges": 16_000_000, "F": 256, "by_negatives": {},
       "source": "scripts/pmc_traffic.sh + scripts/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over "
                 "`python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras [--negatives sigma]`, last full step; "
                 "bytes = 2 x FETCH_SIZE + WRITE_SIZE"}
lines = []
for neg in ("uniform", "sigma"):
    f = json.load(open(f"{src}_{neg}_FETCH_SIZE.json"))["kernels"]
    w = json.load(open(f"{src}_{neg}_WRITE_SIZE.json"))["kernels"]
    lines.append(f"== negatives = {neg}: HBM-side traffic of one step, per kernel (GB; launches in the step)")
    tot = spmm = dec = dec_raw = 0.0
    n_spmm = 0
    for k in f:
        fr, wr = f[k]["sum"] * KIB / 1e9, w.get(k, {"sum": 0.0})["sum"] * KIB / 1e9
        b = 2 * fr + wr
        tot += b
        if k.startswith("spmm_rows_kernel<64"):
            spmm += b
            n_spmm += f[k]["launches"]
        if k.startswith("spmm_long_items_kernel<64"):
            spmm += b
        if any(t in k for t in DEC) and "csr_" not in k:
            dec += b
            dec_raw += fr + wr
        if b >= 0.01:
            lines.append(f"  {k[:72]:72s} n={f[k]['launches']:2d}  fetch {fr:6.3f} (x2)  write {wr:6.3f}  -> {b:6.2f}")
    lines.append(f"  step total {tot:.1f} GB; F=256 aggregation launches {n_spmm}: {spmm / max(n_spmm, 1):.2f} GB per launch; decoder kernels {dec:.2f} GB "
                 f"(fetch counted once: {dec_raw:.2f} GB)")
    out["by_negatives"][neg] = {"traffic_bytes": spmm / max(n_spmm, 1) * 1e9, "decoder_traffic_bytes": dec * 1e9,
                                "decoder_traffic_bytes_fetch_once": dec_raw * 1e9, "step_traffic_bytes": tot * 1e9}
json.dump(out, open(dst + "_traffic.json", "w"), indent=1)
open(dst + "_step_pmc_hbm.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-1:]))
