#!/usr/bin/env python3
"""Regenerates the measurement table of DESIGN.md section 5 from the committed files under profiles/ (bench lines, rocprofv3 kernel
stats, PMC sums) -- every figure in that table is computed here, none is typed by hand.
Usage: tools/design_table.py r03 [--write]     (--write replaces the block between the r-table markers in DESIGN.md)"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(ROOT, "profiles")


def line(name):
    f = os.path.join(P, name)
    if not os.path.exists(f):
        return None
    for l in open(f):
        if l.startswith("{"):
            return json.loads(l)
    return None


def kstats(tag):
    f = os.path.join(P, "%s%s_kernel_stats.csv" % (R, "_" + tag if tag else ""))
    out = {}
    if os.path.exists(f):
        for l in open(f):
            if l.startswith("#") or l.startswith("kernel,"):
                continue
            p = l.rstrip().rsplit(",", 4)
            out[p[0]] = (int(p[1]), float(p[2]), float(p[3]))
    return out


def pmc(tag):
    f = os.path.join(P, "%s%s_pmc_by_kernel.json" % (R, "_" + tag if tag else ""))
    return json.load(open(f)) if os.path.exists(f) else {}


def ratios(v):
    g = lambda n: v.get(n, {}).get("sum", 0.0)
    d = lambda n: max(v.get(n, {}).get("dispatches", 1), 1)
    gui = g("GRBM_GUI_ACTIVE") / d("GRBM_GUI_ACTIVE")   # per launch: the passes are separate runs (the time-based clock spin-up launches a different number of steps in each)
    return {"mfma_busy": (g("SQ_VALU_MFMA_BUSY_CYCLES") / d("SQ_VALU_MFMA_BUSY_CYCLES")) / (gui / 8 * 1024) if gui else None,
            "lds_conflict_share": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None,
            "lds_conflict_per_launch": g("SQ_LDS_BANK_CONFLICT") / d("SQ_LDS_BANK_CONFLICT"),
            "fetch_MB_per_launch": 2 * g("FETCH_SIZE") * 1024 / d("FETCH_SIZE") / 1e6, "write_MB_per_launch": g("WRITE_SIZE") * 1024 / d("WRITE_SIZE") / 1e6}


def find(d, sub):
    for k, v in d.items():
        if sub in k:
            return k, v
    return None, None


rows = []
def add(q, v, src):
    rows.append("| %s | %s | %s |" % (q, v, src))

b = line("%s_bench.json" % R)
if b:
    r = b["roofline"]
    off = b.get("without_kernel_timers")
    add("inference throughput (BASELINE configs[1]: B = 256, N = 1024, fp32, inputs in HBM)", "**%.1f k pairs/s** (%.3f ms/step)%s; sustained %.1f s: %.1f k pairs/s at sclk %s → %s MHz"
        % (b["value"] / 1e3, b["ms_per_step"], "; the same K steps with the per-kernel timers off: %.1f k (%.3f ms)" % (off["value"] / 1e3, off["ms_per_step"]) if off else "",
           b["sustained"]["seconds"], b["sustained"]["value"] / 1e3, b["sustained"]["sclk_mhz_first_chunk"], b["sustained"]["sclk_mhz_last_chunk"]), "`profiles/%s_bench.json`" % R)
    add("dominant kernel", "`%s`, %.0f launches/step, %.1f µs per launch (HIP events in the run)" % (r["kernel"], r["launches_per_step"], r["avg_launch_us"]), "same")
    add("roofline", "%.1f TFLOP/s = **%.3f × %.1f TFLOP/s** fp32 MFMA" % (r["achieved"], r["frac"], r["peak"]), "bench `roofline`")
    k, v = find(pmc(""), "pointnet_fused")
    if v:
        x = ratios(v)
        add("same kernel, PMC", "matrix pipe busy %.3f; LDS bank-conflict share %.4f; HBM %.1f MB fetched (×2-corrected) + %.1f MB written per launch"
            % (x["mfma_busy"], x["lds_conflict_share"] or 0, x["fetch_MB_per_launch"], x["write_MB_per_launch"]), "`profiles/%s_pmc_by_kernel.json`" % R)
    if "infer_bf16x3" in b:
        s = b["infer_bf16x3"]
        add("opt-in split-bf16 backbone (never the headline)", "%.0f k pairs/s (%.2f ms/step), `%s` at %.2f of (bf16 peak / 3)" % (s["value"] / 1e3, s["ms_per_step"], s["roofline"].get("kernel", "pointnet_split"), s["roofline"]["frac"]), "bench `infer_bf16x3`")
    if "pcie_inclusive" in b:
        pc = b["pcie_inclusive"]
        add("host to host, blocking `alignnet_forward` (pageable buffers; the reference's own timing, train.py:447-449)", "%.1f k pairs/s (%.2f ms/step)" % (pc["value"] / 1e3, pc["ms_per_step"]), "bench `pcie_inclusive`; never `value`")
        if "pipelined" in pc:
            add("host to host, pipelined `alignnet_forward_submit / _wait`", "**%.1f k pairs/s** (%.2f ms/step) = %.3f of the device-resident rate" % (pc["pipelined"]["value"] / 1e3, pc["pipelined"]["ms_per_step"], pc["pipelined"]["value"] / b["value"]), "bench `pcie_inclusive.pipelined`")
    if "dgcnn" in b and "infer" in b["dgcnn"]:
        di = b["dgcnn"]["infer"]
        add("short legs of the default line: DGCNN (BASELINE configs[4] shape, N = 4096)", "inference %.2f k pairs/s at %d pairs/step (`%s` at %.3f of the roofline, kNN %.1f ms of %.1f); training at %d pairs/step: fp32 %.2f k, bf16 convs %.2f k pairs/s"
            % (di["value"] / 1e3, di["pairs_per_step"], di["kernel"], di["roofline_frac"] or 0, di["knn_ms_per_step"], di["ms_per_step"], b["dgcnn"]["train_f32"]["pairs_per_step"],
               b["dgcnn"]["train_f32"]["value"] / 1e3, b["dgcnn"]["train_bf16"]["value"] / 1e3), "bench `dgcnn`")
    if "loader" in b and "packed_pairs_per_s" in b["loader"]:
        lo = b["loader"]
        add("short legs: batch loader (SURVEY 8 f1)", "reference-style files %.1f k pairs/s, packed cache %.1f k pairs/s (identical batches: %s); HBM-resident dataset + device sampler feeding fp32 training steps: %.1f k pairs/s"
            % (lo["file_based_pairs_per_s"] / 1e3, lo["packed_pairs_per_s"] / 1e3, lo["identical_batches"], lo.get("device_sampler_train", {}).get("value", 0) / 1e3), "bench `loader`")
    if "icp" in b and "value" in b["icp"]:
        ic = b["icp"]
        add("short legs: ICP refinement (SURVEY 8 f4)", "%.1f k pairs/s (%d pairs x %d points, radius %.1f, <= %d iterations, mean %.1f; fitness %.2f)"
            % (ic["value"] / 1e3, ic["pairs"], ic["points_per_cloud"], ic["radius"], ic["max_iterations"], ic["mean_iterations"], ic["mean_fitness"]), "bench `icp`")
    if "cpu_baseline" in b:
        c = b["cpu_baseline"]
        add("CPU baseline (\"port\": unfused NumPy fp32 oracle, B = 32), thread sweep", ", ".join("%.1f pairs/s on %d" % (x["value"], x["threads"]) for x in c["sweep"]) + " threads (%d host cores)" % c["host_cores"], "bench `cpu_baseline.sweep`")

for tag, name, lab in (("train", "%s_bench_train_f32.json", "training step, fp32"), ("train_bf16", "%s_bench_train_bf16.json", "training step, bf16 convs (BASELINE configs[2])")):
    t = line(name % R)
    if not t:
        continue
    ks = kstats(tag)
    steps = None
    ad = find(ks, "adam_kernel")[1]
    steps = ad[0] if ad else None
    share = t["roofline"]["step_share"]
    off = t.get("without_kernel_timers")
    txt = "**%.1f k pairs/s** (%.3f ms/step%s): " % (t["value"] / 1e3, t["ms_per_step"], "; %.3f ms = %.1f k pairs/s with the per-kernel timers off" % (off["ms_per_step"], off["value"] / 1e3) if off else "") + ", ".join("%s %.2f ms" % (k.replace("train_", ""), v) for k, v in list(share.items())[:4])
    txt += "; dominant `%s` at %.3f of the %s roofline" % (t["roofline"]["kernel"], t["roofline"]["frac"], "bf16-MFMA" if t["roofline"]["peak"] > 1000 else "fp32-MFMA")
    if t["roofline"].get("frac_lift_only") is not None:
        txt += " (%.3f on the lift alone, without the fused Gram's FLOPs)" % t["roofline"]["frac_lift_only"]
    if steps:
        tot = sum(v[0] for v in ks.values()) / steps
        small = sum(v[0] for v in ks.values() if v[2] < 50) / steps
        small_us = sum(v[1] for v in ks.values() if v[2] < 50) / steps
        txt += "; %.0f launches/step, %.0f of them under 50 µs = %.2f ms/step" % (tot, small, small_us / 1e3)
    add(lab, txt, "`profiles/%s`, `profiles/%s_%s_kernel_stats.csv`" % (name % R, R, tag))
    pm = pmc(tag)
    for sub in ("train_fwd_phase3_wide", "train_fwd_phase23<3", "train_bwd_b2", "train_bwd_b1"):
        k, v = find(pm, sub)
        if v:
            x = ratios(v)
            add("&nbsp;&nbsp;`%s` (PMC)" % k.replace("void ", ""), "matrix pipe busy %.3f; LDS bank-conflict share %.3f; HBM %.0f MB fetched + %.0f MB written per launch" % (x["mfma_busy"] or 0, x["lds_conflict_share"] or 0, x["fetch_MB_per_launch"], x["write_MB_per_launch"]), "`profiles/%s_%s_pmc_by_kernel.json`" % (R, tag))

for name, lab in (("%s_bench_dgcnn.json", "DGCNN inference, N = 4096, 512 pairs/step (BASELINE configs[4] shape)"), ("%s_bench_dgcnn_split.json", "same, split-bf16 option")):
    t = line(name % R)
    if t:
        sh = t["roofline"]["step_share"]
        add(lab, "**%.2f k pairs/s** (%.1f ms/step): `%s` %.1f ms = %.3f of the roofline, kNN %.1f ms (%.1f %%)"
            % (t["value"] / 1e3, t["ms_per_step"], t["roofline"]["kernel"], sh.get("backbone", 0), t["roofline"]["frac"], sh.get("knn", 0), 100 * sh.get("knn", 0) / t["ms_per_step"]), "`profiles/%s`" % (name % R))
k, v = find(pmc("dgcnn"), "dgcnn_fused")
if v:
    x = ratios(v)
    add("&nbsp;&nbsp;`dgcnn_fused` (PMC)", "matrix pipe busy %.3f; LDS bank conflicts %.1f M cycles per launch = %.3f of the LDS-active cycles; HBM %.0f MB fetched + %.0f MB written per launch"
        % (x["mfma_busy"] or 0, x["lds_conflict_per_launch"] / 1e6, x["lds_conflict_share"] or 0, x["fetch_MB_per_launch"], x["write_MB_per_launch"]), "`profiles/%s_dgcnn_pmc_by_kernel.json`" % R)
for name, lab, tag, sub in (("%s_bench_train_dgcnn_n1024.json", "DGCNN training, fp32, N = 1024, B = 256", "train_dgcnn", "dg_train_bwd_edge"),
                            ("%s_bench_train_dgcnn_bf16_n1024.json", "DGCNN training, `train_matmul_bf16`, N = 1024, B = 256", "train_dgcnn_bf16", "dg_train_bwd_edge"),
                            ("%s_bench_train_dgcnn_n4096_b64.json", "DGCNN training, fp32, N = 4096, B = 64", None, None),
                            ("%s_bench_train_dgcnn_bf16_n4096_b64.json", "DGCNN training, `train_matmul_bf16`, N = 4096, B = 64", None, None),
                            ("%s_bench_train_dgcnn_n4096_b512.json", "DGCNN training, fp32, N = 4096, B = 512 (BASELINE configs[4]'s per-GPU batch: 1024 clouds fill the 256 CUs, 128 do not)", None, None),
                            ("%s_bench_train_dgcnn_bf16_n4096_b512.json", "DGCNN training, `train_matmul_bf16`, N = 4096, B = 512", None, None)):
    t = line(name % R)
    if not t:
        continue
    sh = t["roofline"]["step_share"]
    add(lab, "**%.2f k pairs/s** (%.1f ms/step): " % (t["value"] / 1e3, t["ms_per_step"]) + ", ".join("%s %.1f" % (k.replace("dg_train_", "edge ").replace("train_", ""), v) for k, v in list(sh.items())[:5]) + " ms", "`profiles/%s`" % (name % R))
    if tag:
        k, v = find(pmc(tag), sub)
        if v:
            x = ratios(v)
            add("&nbsp;&nbsp;`%s` (PMC)" % k.replace("void ", ""), "matrix pipe busy %.3f; LDS bank-conflict share %.3f" % (x["mfma_busy"] or 0, x["lds_conflict_share"] or 0), "`profiles/%s_%s_pmc_by_kernel.json`" % (R, tag))

table = "| quantity | value | source |\n|---|---|---|\n" + "\n".join(rows)
if "--write" in sys.argv:
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    a, z = "<!-- %s-table-begin (tools/design_table.py) -->" % R, "<!-- %s-table-end -->" % R
    i, j = s.index(a) + len(a), s.index(z)
    open(p, "w").write(s[:i] + "\n" + table + "\n" + s[j:])
    print("DESIGN.md updated (%d rows)" % len(rows))
else:
    print(table)
