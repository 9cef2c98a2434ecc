#!/usr/bin/env python3
"""Condense the rocprofv3 output of profiles/run_profiles_r02.sh into the two committed artefacts:

    profiles/<tag>_summary.txt   human-readable: kernel-trace stats, per-launch counters of the dominant kernels, derived figures
    profiles/pmc_r02.json        what bench.py reads for roofline.traffic / roofline.pmc (per-launch averages, same command)

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md ("HBM" / rocprofv3): FETCH_SIZE and WRITE_SIZE come from separate --pmc passes
(kernel-trace only), are in KiB, and on gfx950 FETCH_SIZE tallies wide coalesced reads at half their size -- the factor is MEASURED in the same
run (torch reduction over a 1 GiB buffer, calib_fetch/) and applied.  Formulas used (DESIGN.md section 7 repeats them):
    hbm_bytes_per_launch  = FETCH_SIZE[KiB] * 1024 * read_factor + WRITE_SIZE[KiB] * 1024
    l2_read_bytes         = TCC_REQ_sum * 128          (one request = one 128-byte line)
    valu cycles / inst    = kernel_avg_us * 1e-6 * 2.4e9 * 1024 SIMDs / SQ_INSTS_VALU     (how long a SIMD spends per VALU wave-instruction
                            if it did nothing else: ~4 = issue-bound on gfx950's 16-lane SIMDs)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

SIMDS = 256 * 4
CLOCK_HZ = 2.4e9


def read_counters(path):
    """{kernel name: {counter: average per dispatch}} (the per-XCD / per-SE rows of one dispatch are summed)"""
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (d, c), v in per.items():
            out[names[d]][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}


def kernel_stats(path):
    rows = {}
    for f in glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"]))
    return rows


def find(d, sub):
    for k in d:
        if sub in k:
            return k
    return None


def main():
    d, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = [f"# rocprofv3 summary {tag}  (source: profiles/run_profiles_{tag[:3]}.sh; raw CSVs stay in gpurun_out/)"]
    bench = None
    try:
        bench = json.loads([l for l in open(os.path.join(d, "trace.json")).read().strip().splitlines() if l.startswith("{")][-1])  # (RCCL prints a banner on stdout too)
        lines.append(f"bench line of the traced run: value={bench['value']:.4g} {bench['unit']}  ms_per_step={bench['ms_per_step']:.4f}  "
                     f"kernel_ms={bench['roofline']['kernel_ms']:.4f}")
    except Exception as e:
        lines.append(f"(bench line of the traced run unreadable: {e})")
    stats = kernel_stats(os.path.join(d, "trace"))
    lines.append("\n## kernel-trace --stats (our kernels): calls, average / min / max us, share")
    for k, (c, a, mn, mx, pct) in sorted(stats.items(), key=lambda kv: -kv[1][4]):
        if "gsh::" in k:
            lines.append(f"{k.replace('void ', '').replace('gsh::(anonymous namespace)::', '').split('(')[0][:70]:70s} calls={c:5d} avg={a:9.2f} min={mn:9.2f} max={mx:9.2f} {pct:5.1f}%")
    # calibration of FETCH_SIZE on wide reads
    rf = 2.0
    cal = read_counters(os.path.join(d, "calib_fetch"))
    k = find(cal, "reduce_kernel")
    if k and cal[k].get("FETCH_SIZE"):
        rf = (1 << 30) / (cal[k]["FETCH_SIZE"] * 1024.0)
        lines.append(f"\ncalibration: torch sum over 1 GiB -> FETCH_SIZE {cal[k]['FETCH_SIZE']:.0f} KiB per launch -> read_factor = {rf:.3f}")
    legs = {name: read_counters(os.path.join(d, name)) for name in ("fetch", "write", "sq1", "sq2", "tcc")}
    out = {}
    for label, sub, extra in (("mcorr", "mcorr_kernel<3, 0, false", {}), ("oc_cell", "oc_cell_kernel", {}), ("oc_forward", "oc_forward_kernel", {}),
                              ("trk_loop", "trk_loop_kernel<3, false, false>", {}), ("oc_subcell_dit", "oc_subcell_dit_kernel", {}), ("oc_combine_dit", "oc_combine_dit_kernel", {})):
        rec = {}
        if label == "mcorr" and find(stats, "mcorr_kernel_t128<3, 0, false"):
            sub = "mcorr_kernel_t128<3, 0, false"   # round 6: launches of >= 5 120 E/P/L jobs run the two-wave kernels (csrc/multicorrelator_t128.hip)
        for leg in legs.values():
            kk = find(leg, sub)
            if kk:
                rec.update(leg[kk])
        ks = find(stats, sub)
        if ks:
            rec["kernel_avg_us"] = stats[ks][1]
            rec["kernel_calls"] = stats[ks][0]
        if not rec:
            continue
        if "FETCH_SIZE" in rec or "WRITE_SIZE" in rec:
            rec["hbm_bytes_per_launch"] = rec.get("FETCH_SIZE", 0.0) * 1024.0 * rf + rec.get("WRITE_SIZE", 0.0) * 1024.0
        if "TCC_REQ_sum" in rec:
            rec["l2_read_bytes_per_launch"] = rec["TCC_REQ_sum"] * 128.0
        if "kernel_avg_us" in rec:
            t = rec["kernel_avg_us"] * 1e-6
            if "hbm_bytes_per_launch" in rec:
                rec["hbm_GBs"] = rec["hbm_bytes_per_launch"] / t / 1e9
            if "l2_read_bytes_per_launch" in rec:
                rec["l2_GBs"] = rec["l2_read_bytes_per_launch"] / t / 1e9
            if rec.get("SQ_INSTS_VALU"):
                rec["valu_cycles_per_inst_per_simd"] = t * CLOCK_HZ * SIMDS / rec["SQ_INSTS_VALU"]
        rec["source"] = f"profiles/{tag}_summary.txt (gpurun_out/prof_{tag})"
        out[label] = rec
        lines.append(f"\n## {label}: {sub} -- averages per launch")
        for c in sorted(rec):
            if c != "source":
                lines.append(f"  {c:36s} {rec[c]:.6g}" if isinstance(rec[c], float) else f"  {c:36s} {rec[c]}")
    # shape keys bench.py matches against
    if bench:
        cfg = bench.get("config", {})
        if "mcorr" in out:
            out["mcorr"]["jobs"] = cfg.get("channels_per_gpu", 0) * cfg.get("epochs_per_block", 0)
            out["mcorr"]["n"] = cfg.get("samples_per_epoch")
            cs = float(out["mcorr"]["jobs"]) * float(out["mcorr"]["n"] or 0)
            if cs and out["mcorr"].get("SQ_INSTS_VALU"):
                out["mcorr"]["valu_insts_per_channel_sample"] = out["mcorr"]["SQ_INSTS_VALU"] * 64.0 / cs
        if "oc_cell" in out:
            acq = {"n": 25000, "n_prn": 32, "n_bins": 41, "source": out["oc_cell"]["source"]}
            hb = out["oc_cell"].get("hbm_bytes_per_launch", 0.0) + out.get("oc_forward", {}).get("hbm_bytes_per_launch", 0.0)
            acq["hbm_bytes_per_batch"] = hb
            for k2 in ("SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "kernel_avg_us", "l2_read_bytes_per_launch"):
                if k2 in out["oc_cell"]:
                    acq["cell_" + k2] = out["oc_cell"][k2]
            out["acquisition"] = acq
    lines += acquisition_channel_traces(d)
    open(os.path.join(root, "profiles", f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(root, "profiles", os.environ.get("GSH_PMC_JSON", "pmc_r02.json")), "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


def acquisition_channel_traces(prof):
    """Round 3: kernel traces of eight acquisition channels on one stream, through the shared runtime and on their own handles (tests/host/test_adapters
    acq_shared | acq_alone).  Code transforms (set_local_code: one work-group) and signal transforms (one work-group per Doppler bin) are the same kernel and
    are told apart by the grid."""
    import collections
    import re
    lines = []
    for mode in ("acq_shared", "acq_alone"):
        files = glob.glob(os.path.join(prof, mode, "**", "*kernel_trace.csv"), recursive=True)
        if not files:
            continue
        if not lines:
            lines.append("\n## eight acquisition channels on one stream (tests/host/test_adapters acq_shared / acq_alone): launches of the on-chip kernels")
        said = [ln.strip() for ln in open(os.path.join(prof, mode + ".log"), errors="replace") if ln.startswith(mode + ":")] if os.path.exists(os.path.join(prof, mode + ".log")) else []
        lines.append(f"# {said[-1] if said else mode}")
        c = collections.Counter()
        for r in csv.DictReader(open(files[0])):
            m = re.search(r"(oc_\w+)<", r["Kernel_Name"])
            if m:
                c[(m.group(1), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))] += 1
        for (name, wgs), n in sorted(c.items()):
            what = ("code spectrum (set_local_code)" if wgs == 1 else "signal spectra, one work-group per bin") if name == "oc_forward_kernel" else "cells (PRNs x bins)"
            lines.append("   %-20s %5d work-groups  x %3d launches   %s" % (name, wgs, n, what))
    return lines


if __name__ == "__main__":
    main()
