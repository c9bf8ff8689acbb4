"""Summarise the rocprofv3 runs of tools/run_profiles.sh into the committed files under profiles/.

usage: python3 tools/profile_summarize.py <prof_dir> <out_dir> <tag>
  <prof_dir>/stats  : --kernel-trace --stats            -> <tag>_bench_kernel_stats.{csv,md}
  <prof_dir>/fetch, write : --pmc FETCH_SIZE / WRITE_SIZE -> <tag>_pmc_traffic.{json,md}
  <prof_dir>/mfma   : --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -> <tag>_pmc_mfma.md
gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads -> doubled; WRITE_SIZE exact for 16-B/lane stores; both in units the CSV reports (bytes here: the
counter values are multiplied by the unit factor printed below if rocprofv3 reports KiB)."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def find(d, pat):
    f = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)   # the newest run's file
    return f[-1] if f else None


def short(name):
    return name if len(name) <= 100 else name[:97] + "..."


KIB = 1024.0


def gemm_family(name):
    """gemm_dma_kernel<E, A_KMAJOR, B_KMAJOR, CFG, EPI, OUT, A_CONV> -> fwd (k,k) / dgrad (k,mn) / wgrad (mn,mn).
    rocprofv3 leaves some instantiations mangled (IDF16bLb<AK>ELb<BK>E...) and mis-demangles the bf16 argument of
    the others as 'bool _Accum, bool, E, <BK>, ...' (A_KMAJOR is swallowed; only wgrad has A_KMAJOR = false and that
    form stays mangled)."""
    if "gemm_dma_kernel" not in name:
        return None
    m = re.search(r"gemm_dma_kernelIDF16[b_]Lb(\d)ELb(\d)E", name)
    if m:
        ak, bk = m.group(1) == "1", m.group(2) == "1"
    else:
        m = re.search(r"E, (true|false), \d", name)
        if not m:
            return None
        ak, bk = True, m.group(1) == "true"
    return "fwd" if (ak and bk) else "dgrad" if ak else "wgrad"


def conv_family(name):
    """Kernels of the per-frame CNN encoder: the implicit-GEMM convolution (gemm_dma_kernel with A_CONV = true: forward /
    data gradient when A is k-major, weight gradient when both operands are mn-major) and the LDS-halo 3x3 kernel."""
    if "conv3x3_stream_kernel" in name:
        # MODE 1 / 2 of the (3, 1) form: the two passes of the fused data gradient + BatchNorm backward (one operator = one
        # launch of each; bench.py doubles the family's per-launch average)
        if re.search(r"Li64ELi144ELi3ELi[12]E", name) or re.search(r"64, 144, 3, [12]>", name):
            return "conv3x1_stream_bn_bwd"
        return "conv3x3_stream"
    if "conv3x3_c64_wgrad_kernel" in name:
        return "conv3x3_c64_wgrad"
    if "conv3x1_dbn_kernel" in name:                       # round 6: the two passes of the same operator as a window kernel
        return "conv3x1_stream_bn_bwd"
    if "conv3x1_wgrad_kernel" in name or "conv3x1_wgrad_pipe_kernel" in name:
        return "conv3x1_wgrad"
    if "conv3x1_fwd_kernel" in name or "conv3x1_fwd_pipe_kernel" in name:
        return "conv3x1_fwd"
    if "conv3x1_c64_kernel" in name:
        return "conv3x1_c64"
    if "conv_stem_kernel" in name:
        return "conv_stem7"
    if "conv3x3_c64_kernel" in name:
        return "conv3x3_c64"
    if "gemm_dma_kernel" not in name:
        return None
    m = re.search(r"gemm_dma_kernelIDF16[b_]Lb(\d)ELb(\d)ELi\d+ELi\d+ELi\d+ELb(\d)E", name)
    if m:
        ak, conv = m.group(1) == "1", m.group(3) == "1"
    else:
        m = re.search(r"E, (true|false), \d+, \d+, \d+, (true|false)(?:, (?:true|false))?>", name)   # (+ the BNB flag of round 4)
        if not m:
            return None
        ak, conv = True, m.group(2) == "true"
    if not conv:
        return None
    return "conv_implicit" if ak else "conv_wgrad"


def traffic_tables(prof, sub_f, sub_w, outd, stem, title, family_of):
    """FETCH_SIZE / WRITE_SIZE passes -> <stem>.md (per kernel and grid) and <stem>.json (per family)."""
    fe, wr = load_counters(os.path.join(prof, sub_f)), load_counters(os.path.join(prof, sub_w))
    if not (fe or wr):
        return
    keys = sorted(set(fe) | set(wr), key=lambda k: -(2 * fe[k].get("FETCH_SIZE", 0) + wr[k].get("WRITE_SIZE", 0)))
    fam = {}
    with open(os.path.join(outd, stem + ".md"), "w") as fh:
        fh.write(title + "| kernel | grid | launches | FETCH raw | WRITE | corrected 2F+W |\n|---|---|---|---|---|---|\n")
        for k in keys[:48]:
            n = max(fe[k].get("n", 0), wr[k].get("n", 0), 1)
            f_mb = fe[k].get("FETCH_SIZE", 0) * KIB / n / 1e6      # rocprofv3 reports both counters in KiB
            w_mb = wr[k].get("WRITE_SIZE", 0) * KIB / n / 1e6
            fh.write(f"| `{short(k[0])}` | {k[1]} | {int(n)} | {f_mb:.1f} | {w_mb:.1f} | {2*f_mb+w_mb:.1f} |\n")
        for k in keys:
            n = max(fe[k].get("n", 0), wr[k].get("n", 0), 1)
            f_mb = fe[k].get("FETCH_SIZE", 0) * KIB / n / 1e6
            w_mb = wr[k].get("WRITE_SIZE", 0) * KIB / n / 1e6
            famname = family_of(k[0])
            if famname:
                a = fam.setdefault(famname, [0.0, 0])
                a[0] += (2 * f_mb + w_mb) * 1e6 * n
                a[1] += n
    json.dump({"families": {k: {"hbm_bytes_corrected": int(v[0] / max(v[1], 1)), "launches": int(v[1])} for k, v in fam.items()},
               "note": "bytes per launch, 2*FETCH_SIZE + WRITE_SIZE, averaged over the family's launches"},
              open(os.path.join(outd, stem + ".json"), "w"), indent=1)


def load_counters(d):
    """-> {(kernel, grid): {"n": launches, counter: sum}}"""
    f = find(d, "*counter_collection.csv")
    out = defaultdict(lambda: defaultdict(float))
    if not f:
        return out
    seen = set()
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], int(r["Grid_Size"]))
        out[key][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (r["Dispatch_Id"], r["Counter_Name"])
        if r["Counter_Name"] and (r["Dispatch_Id"],) not in seen:
            seen.add((r["Dispatch_Id"],))
            out[key]["n"] += 1
    return out


def main():
    prof, outd, tag = sys.argv[1:4]
    os.makedirs(outd, exist_ok=True)
    # ---- kernel stats
    st = find(os.path.join(prof, "stats"), "*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(outd, f"{tag}_bench_kernel_stats.csv"))
        rows = list(csv.DictReader(open(st)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        with open(os.path.join(outd, f"{tag}_bench_kernel_stats.md"), "w") as fh:
            fh.write(f"# rocprofv3 --kernel-trace --stats ({tag})\n\nCommand: tools/run_profiles.sh (bench.py --steps 5 --warmup 2, "
                     "hipGraph replay: 5 timed + 2 warm-up + 2 capture warm-up steps = 9 steps in the trace).\n\n"
                     "| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
            for r in rows[:32]:
                fh.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                         f"{float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |\n")
            fh.write(f"\nTotal kernel time {tot/1e6:.2f} ms over 9 steps = {tot/9e6:.2f} ms/step.\n")
    # ---- the same trace split by (kernel, grid): rocprofv3's per-name averages mix the 50,432-row space launches with the
    #      264-row temporal ones
    tr = find(os.path.join(prof, "stats"), "*kernel_trace.csv")
    if tr:
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(tr)):
            gs = int(r.get("Grid_Size") or int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
            ws = int(r.get("Workgroup_Size") or int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
            k = (r["Kernel_Name"], gs, ws)
            agg[k][0] += 1
            agg[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        tot = sum(v[1] for v in agg.values())
        with open(os.path.join(outd, f"{tag}_bench_kernel_by_grid.md"), "w") as fh:
            fh.write(f"# kernel trace split by (kernel, grid, block) ({tag})\n\n9 steps in the trace; calls and ms are per step.\n\n"
                     "| kernel | grid | block | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|---|---|\n")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
                fh.write(f"| `{short(k[0])}` | {k[1]} | {k[2]} | {v[0]/9:.2f} | {v[1]/9e6:.4f} | {v[1]/v[0]/1e3:.1f} | "
                         f"{100*v[1]/tot:.1f} |\n")
            fh.write(f"\nTotal kernel time {tot/9e6:.3f} ms/step.\n")
    # Per-STEP tables (all workloads): a step = the launches between two optimizer launches, so that one-time set-up work
    # (parameter copies into the flat buffers, warm-up allocations: the "copyBuffer per step" of the r03 tables was that,
    # divided by the step count) does not appear as step work.  hipGraph replay, as the bench line is measured.
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev"))
    import trace_steps
    for wl, sub in (("bench", "stats"), ("pyramid", "stats_pyramid"), ("frametransformer", "stats_frametransformer"),
                    ("longclip", "stats_longclip")):
        tr2 = find(os.path.join(prof, sub), "*kernel_trace.csv")
        if not tr2:
            continue
        name = f"{tag}_{wl}_kernel_stats.md" if wl != "bench" else f"{tag}_bench_kernel_per_step.md"
        with open(os.path.join(outd, name), "w") as fh:
            fh.write(f"# rocprofv3 --kernel-trace, bench.py" + (f" --workload {wl}" if wl != "bench" else "") + f" ({tag}): per-step table\n\n"
                     "Steps = intervals between consecutive `adamw_fused_kernel` launches (hipGraph replay); the last 3 averaged.\n\n")
            trace_steps.table(tr2, 3, 260, fh, os.path.join(outd, f"{tag}_{wl}_last_step_order.txt") if wl == "bench" else None)
    # ---- traffic
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), {cmd}.  gfx950: FETCH_SIZE counts "
            "64 B per 128-B request for wide coalesced reads -> doubled; WRITE_SIZE exact.  FETCH_SIZE is counted at the L2 fabric side "
            "and includes Infinity-Cache hits (upper bound of HBM reads).  Values: MB per launch.\n\n")
    traffic_tables(prof, "fetch", "write", outd, f"{tag}_pmc_traffic", f"# HBM traffic per launch from PMC counters ({tag})\n\n"
                   + note.format(cmd="bench.py --steps 2 --warmup 1 --no-graph"), gemm_family)
    traffic_tables(prof, "fetch_pyramid", "write_pyramid", outd, f"{tag}_pyramid_pmc_traffic",
                   f"# HBM traffic per launch from PMC counters, bench.py --workload pyramid ({tag})\n\n"
                   + note.format(cmd="bench.py --workload pyramid --steps 1 --warmup 1 --no-graph"), conv_family)
    traffic_tables(prof, "fetch_frametransformer", "write_frametransformer", outd, f"{tag}_frametransformer_pmc_traffic",
                   f"# HBM traffic per launch from PMC counters, bench.py --workload frametransformer ({tag})\n\n"
                   + note.format(cmd="bench.py --workload frametransformer --steps 1 --warmup 1 --no-graph"), conv_family)
    # ---- MFMA busy
    mf = load_counters(os.path.join(prof, "mfma"))
    if mf:
        keys = sorted(mf, key=lambda k: -mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
        with open(os.path.join(outd, f"{tag}_pmc_mfma.md"), "w") as fh:
            fh.write(f"# MFMA pipe utilisation from PMC counters ({tag})\n\nrocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                     "(--kernel-trace only).  util = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): the share of "
                     "SIMD-cycles in which a matrix instruction occupies the pipe.\n\n| kernel | grid | launches | MFMA busy Mcyc | "
                     "GUI active Mcyc/XCD | util |\n|---|---|---|---|---|---|\n")
            for k in keys[:24]:
                b, g = mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0), mf[k].get("GRBM_GUI_ACTIVE", 0) / 8
                if b <= 0 or g <= 0:
                    continue
                fh.write(f"| `{short(k[0])}` | {k[1]} | {int(mf[k]['n'])} | {b/1e6:.1f} | {g/1e6:.2f} | {b/(g*1024):.3f} |\n")
    print("wrote", sorted(os.listdir(outd)))


if __name__ == "__main__":
    main()
