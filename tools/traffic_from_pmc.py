#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes into profiles/traffic_<workload>.json, the file bench.py reads for
`roofline.traffic` (HBM-side bytes per launch).

Method = MI355X_MICROARCH.md, section "HBM": FETCH_SIZE and WRITE_SIZE come from the L2's
memory-side request counters and need SEPARATE passes (TCC has 4 slots: FETCH_SIZE takes 3,
WRITE_SIZE 2); both are reported in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request of a
16-byte-per-lane stream as 64 bytes, so it is doubled -- and that factor is re-checked in the
same run on a launch whose read volume is known exactly (`unique` pattern of
tools/profile_forward.py: B*H distinct rows, every row read once).  Infinity-Cache hits are
inside these counters: the figure is "bytes that crossed the L2 -> fabric boundary", an upper
bound on HBM traffic.

    python tools/traffic_from_pmc.py --forward-fetch DIR --forward-write DIR [--forward-tcc DIR]
           [--pipeline-fetch DIR --pipeline-write DIR [--pipeline-tcc DIR] [--pipeline-trace DIR]]
           --iters N --out profiles/traffic_c2.json

The forward passes are runs of `tools/profile_forward.py --pattern all --iters N` (N launches of
the power-law C2 batch, then N of the uniform one, then N of the unique-rows one); the pipeline
passes are runs of `benchmarks/manual_benchmark <C2 flags>` (forward, transpose, backward).
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FETCH_CORRECTION = 2.0
WRITE_CORRECTION = 1.0
SORT_FAMILY = ("Radix", "RunHead", "FillQuotient", "SingleTileSort", "ExpandCsr")


def read_counters(d):
    """[(dispatch order, kernel name, {counter: value})] of one rocprofv3 --pmc output directory."""
    per = defaultdict(dict)
    names = {}
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = int(r["Dispatch_Id"])
                per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names[k] = r["Kernel_Name"]
    return [(k, names[k], per[k]) for k in sorted(per)]


def read_trace(d):
    out = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                out[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


def avg(v):
    return sum(v) / len(v) if v else None


def pick(disp, substr, counter):
    return [c[counter] for _, n, c in disp if substr in n and counter in c]


def entry(fetch_kb, write_kb, launches, kernel):
    e = {"kernel": kernel, "launches_averaged": launches,
         "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb}
    if fetch_kb is not None and write_kb is not None:
        e["hbm_bytes_per_launch"] = int(round((fetch_kb * FETCH_CORRECTION + write_kb * WRITE_CORRECTION) * 1024))
    return e


def main():
    p = argparse.ArgumentParser()
    for k in ("forward-fetch", "forward-write", "forward-tcc", "pipeline-fetch", "pipeline-write",
              "pipeline-tcc", "pipeline-trace", "c3-fetch", "c3-write", "c3-tcc", "blocks-fetch", "blocks-write", "blocks-tcc",
              "coalesced-fetch", "coalesced-write", "coalesced-tcc"):
        p.add_argument("--" + k)
    p.add_argument("--coalesced-launches-per-call", type=int, default=2)
    p.add_argument("--iters", type=int, default=0, help="launches per pattern in the forward passes")
    p.add_argument("--expected-unique-read-bytes", type=int, default=65536 * 64 * 512)
    p.add_argument("--workload", default="c2 (fp16 sum, 10Mx256, batch 65536, hotness 64, alpha 1.15) + its C4 backward")
    p.add_argument("--out", required=True)
    p.add_argument("--rows-out", help="text file: one `traffic_row` line per entry of the JSON (the per-pass averages "
                                      "it was computed from), committed under profiles/ next to the pass summaries; "
                                      "tests/test_bench_contract.py recomputes the JSON's byte counts from it")
    a = p.parse_args()
    res = {"workload": a.workload,
           "generated_by": "tools/traffic_from_pmc.py from the rocprofv3 --pmc passes of tools/refresh_profiles.sh",
           "method": "FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (KiB per dispatch, averaged over "
                     "the launches of the kernel); hbm_bytes = (FETCH_SIZE * %.1f + WRITE_SIZE * %.1f) * 1024; the "
                     "FETCH_SIZE factor is the gfx950 correction of MI355X_MICROARCH.md (128-B requests tallied as "
                     "64 B), re-checked in `calibration_unique_rows`; L2->fabric requests, Infinity-Cache hits included"
                     % (FETCH_CORRECTION, WRITE_CORRECTION),
           "fetch_correction": FETCH_CORRECTION, "write_correction": WRITE_CORRECTION, "kernels": {}}
    import bench
    res["kernel_sources_sha16"] = bench.kernel_sources_sha16()   # bench.py reports a mismatch as traffic_stale
    K = res["kernels"]
    if a.c3_fetch and a.c3_write:
        # passes over `bench.py --workload c3 --no-extras --no-cpu-baseline`: every GatherReduceKernel dispatch
        # is a C3 forward launch (two CSR batches cycled); the first launches run on cold caches: drop two
        cf = pick(read_counters(a.c3_fetch), "GatherReduceKernel", "FETCH_SIZE")
        cw = pick(read_counters(a.c3_write), "GatherReduceKernel", "WRITE_SIZE")
        K["forward_c3"] = entry(avg(cf[2:]), avg(cw[2:]), len(cf) - 2, "GatherReduceKernel")
        if a.c3_tcc:
            t = read_counters(a.c3_tcc)
            h, m = avg(pick(t, "GatherReduceKernel", "TCC_HIT_sum")[2:]), avg(pick(t, "GatherReduceKernel", "TCC_MISS_sum")[2:])
            K["forward_c3"]["l2"] = {"TCC_HIT_sum": h, "TCC_MISS_sum": m, "hit_rate": round(h / (h + m), 4)}
    if a.forward_fetch and a.forward_write:
        ff = pick(read_counters(a.forward_fetch), "GatherReduceKernel", "FETCH_SIZE")
        fw = pick(read_counters(a.forward_write), "GatherReduceKernel", "WRITE_SIZE")
        n = a.iters
        assert len(ff) == 3 * n and len(fw) == 3 * n, (len(ff), len(fw), n)
        for i, name in enumerate(("forward_c2", "forward_c2_alpha0", "calibration_unique_rows")):
            # the first launch of each pattern runs on caches warmed by another pattern: drop it
            K[name] = entry(avg(ff[i * n + 1:(i + 1) * n]), avg(fw[i * n + 1:(i + 1) * n]), n - 1, "GatherReduceKernel")
        cal = K["calibration_unique_rows"]
        cal["expected_read_bytes"] = a.expected_unique_read_bytes + 65536 * 64 * 4  # rows + the index stream
        cal["measured_read_bytes"] = int(round(cal["fetch_size_kb_per_launch"] * FETCH_CORRECTION * 1024))
        cal["measured_over_expected"] = round(cal["measured_read_bytes"] / cal["expected_read_bytes"], 4)
        if a.forward_tcc:
            t = read_counters(a.forward_tcc)
            hit, miss = pick(t, "GatherReduceKernel", "TCC_HIT_sum"), pick(t, "GatherReduceKernel", "TCC_MISS_sum")
            for i, name in enumerate(("forward_c2", "forward_c2_alpha0", "calibration_unique_rows")):
                h, m = avg(hit[i * n + 1:(i + 1) * n]), avg(miss[i * n + 1:(i + 1) * n])
                K[name]["l2"] = {"TCC_HIT_sum": h, "TCC_MISS_sum": m, "hit_rate": round(h / (h + m), 4)}
    if a.pipeline_fetch and a.pipeline_write:
        pf, pw = read_counters(a.pipeline_fetch), read_counters(a.pipeline_write)
        bf = pick(pf, "SegmentedScatterAddKernel", "FETCH_SIZE")
        bw = pick(pw, "SegmentedScatterAddKernel", "WRITE_SIZE")
        K["backward_c4"] = entry(avg(bf), avg(bw), len(bf), "SegmentedScatterAddKernel")
        calls = max(len(bf), 1)     # one transpose (+ remap) per backward in the benchmark loop
        sf = [c["FETCH_SIZE"] for _, n_, c in pf if any(s in n_ for s in SORT_FAMILY) and "FETCH_SIZE" in c]
        sw = [c["WRITE_SIZE"] for _, n_, c in pw if any(s in n_ for s in SORT_FAMILY) and "WRITE_SIZE" in c]
        K["transpose_c4"] = entry(sum(sf) / calls, sum(sw) / calls, calls,
                                  "row ids + radix sort + run-head scan kernels of one Transpose/remap call")
        K["transpose_c4"]["kernel_launches_per_call"] = round(len(sf) / calls, 2)
        if a.pipeline_tcc:
            t = read_counters(a.pipeline_tcc)
            h, m = avg(pick(t, "SegmentedScatterAddKernel", "TCC_HIT_sum")), avg(pick(t, "SegmentedScatterAddKernel", "TCC_MISS_sum"))
            K["backward_c4"]["l2"] = {"TCC_HIT_sum": h, "TCC_MISS_sum": m, "hit_rate": round(h / (h + m), 4)}
        if a.pipeline_trace:
            tr = read_trace(a.pipeline_trace)
            for name, v in tr.items():
                if "SegmentedScatterAddKernel" in name:
                    K["backward_c4"]["kernel_ms_profiled"] = round(avg(v) / 1e6, 5)
    if a.blocks_fetch and a.blocks_write:
        # the same pipeline with Transpose(sample_blocks = recommended): the backward's traffic on the blocked order
        bf = pick(read_counters(a.blocks_fetch), "SegmentedScatterAddKernel", "FETCH_SIZE")
        bw = pick(read_counters(a.blocks_write), "SegmentedScatterAddKernel", "WRITE_SIZE")
        K["backward_c4_sample_blocks"] = entry(avg(bf), avg(bw), len(bf), "SegmentedScatterAddKernel")
        if a.blocks_tcc:
            tb = read_counters(a.blocks_tcc)
            h, m = avg(pick(tb, "SegmentedScatterAddKernel", "TCC_HIT_sum")), avg(pick(tb, "SegmentedScatterAddKernel", "TCC_MISS_sum"))
            K["backward_c4_sample_blocks"]["l2"] = {"TCC_HIT_sum": h, "TCC_MISS_sum": m, "hit_rate": round(h / (h + m), 4)}
    if a.coalesced_fetch and a.coalesced_write:
        # the blocked order with the REFERENCE's compressed gradient (--coalesce_blocks): one scatter launch per sample
        # block; the bytes of one EmbeddingBackward call are those of its launches together
        cf = pick(read_counters(a.coalesced_fetch), "SegmentedScatterAddKernel", "FETCH_SIZE")
        cw = pick(read_counters(a.coalesced_write), "SegmentedScatterAddKernel", "WRITE_SIZE")
        per_call = a.coalesced_launches_per_call
        calls = max(len(cf) // per_call, 1)
        K["backward_c4_blocked_coalesced"] = entry(sum(cf) / calls, sum(cw) / calls, calls,
                                                   "SegmentedScatterAddKernel x %d (one per sample block)" % per_call)
        if a.coalesced_tcc:
            tb = read_counters(a.coalesced_tcc)
            h, m = sum(pick(tb, "SegmentedScatterAddKernel", "TCC_HIT_sum")), sum(pick(tb, "SegmentedScatterAddKernel", "TCC_MISS_sum"))
            K["backward_c4_blocked_coalesced"]["l2"] = {"TCC_HIT_sum": h / calls, "TCC_MISS_sum": m / calls,
                                                        "hit_rate": round(h / (h + m), 4)}
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    if a.rows_out:
        with open(a.rows_out, "a") as f:          # (from the entries themselves: no reliance on creation order)
            for name, e in K.items():
                if not isinstance(e, dict) or e.get("fetch_size_kb_per_launch") is None:
                    continue
                f.write("traffic_row %s %s FETCH_SIZE_KiB=%.4f WRITE_SIZE_KiB=%.4f dispatches=%d kernel=%s\n"
                        % (os.path.basename(a.out), name, e["fetch_size_kb_per_launch"], e["write_size_kb_per_launch"],
                           e["launches_averaged"], e["kernel"].replace(" ", "_")))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
