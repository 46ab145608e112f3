#!/usr/bin/env python3
"""Condense rocprofv3 rocpd databases (gpurun_out/prof_<tag>) into small text files for profiles/.
Usage: tools/summarize_prof.py gpurun_out/prof_r01 r01 [dest_dir]"""
import re, glob, json, os, sqlite3, sys
from collections import defaultdict

if __name__ != "__main__":   # imported by tests/test_bench_launcher_cpu.py for traffic_summary(): nothing to read
    out = tag = dst = None
else:
    out, tag = sys.argv[1], sys.argv[2]
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(out, "summary")
    os.makedirs(dst, exist_ok=True)

def short(n):
    n = n.split("(")[0].replace("alignnet::", "")
    # instantiations that are ONE kernel to bench.py (its kernel timers and `roofline.launches_per_step` count them together): the
    # persistent split-bf16 backbone runs as <4> (C3 >= 512) and <2> (C3 = 256) within one step
    return re.sub(r"pointnet_split_persist<\d+>", "pointnet_split_persist", n)

# 1. kernel-trace stats (rocprofv3 --kernel-trace --stats)
for db in (glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True) if out else []):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as g:
        g.write("# rocprofv3 --kernel-trace --stats -- python bench.py (durations in microseconds)\n")
        g.write("kernel,calls,total_us,average_us,percent\n")
        for n, calls, tot, avg, pct in rows:
            g.write(f"{short(n)},{calls},{tot:.3f},{avg:.3f},{pct:.3f}\n")
    # per-launch detail of the dominant kernel (three launch shapes per step)
    det = defaultdict(list)
    try:
        for name, gx, dur in c.execute("select name, grid_x*grid_y*grid_z, (end-start) from kernels"):
            det[(short(name), gx)].append(dur)
        with open(os.path.join(dst, f"{tag}_kernel_by_grid.csv"), "w") as g:
            g.write("kernel,grid_size,calls,average_us,min_us,max_us\n")
            for (n, gx), v in sorted(det.items(), key=lambda kv: -sum(kv[1])):
                g.write(f"{n},{gx},{len(v)},{sum(v)/len(v)/1e3:.3f},{min(v)/1e3:.3f},{max(v)/1e3:.3f}\n")
    except sqlite3.Error as e:
        print("kernels view:", e)
    break

# 2. PMC passes (each in its own run, no tracing flags)
pmc = defaultdict(dict)
for db in (glob.glob(os.path.join(out, "pmc_*", "**", "*.db"), recursive=True) if out else []):
    c = sqlite3.connect(db)
    acc, cnt = defaultdict(float), defaultdict(set)
    for name, ctr, val, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        acc[(short(name), ctr)] += val; cnt[(short(name), ctr)].add(did)
    for (k, ctr), v in acc.items():
        pmc[k][ctr] = {"sum": v, "dispatches": len(cnt[(k, ctr)])}
if out:
    json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_by_kernel.json"), "w"), indent=1, sort_keys=True)

# 3. HBM traffic per launch and per bench step.  FETCH_SIZE / WRITE_SIZE are KiB.  gfx950: FETCH_SIZE tallies a 128-B
#    request as 64 B on wide coalesced streams (MI355X_MICROARCH.md, HBM) -> x2 on the read side.
#    Every figure of a pass is normalised by THAT pass's own launch / step counts (the passes are separate runs whose time-based
#    spin-up makes different numbers of steps; round 3 divided the PMC passes' bytes by the trace run's step count).
def steps_of(log):
    """every step the profiled process ran, from the bench line it printed (one leg per profile run), and the leg's shape"""
    try:
        for line in open(log):
            if line.startswith("{"):
                j = json.loads(line)
                # bench.py counts every call of the headline step itself ("steps_executed": spin-up, warm-up, the K timed steps, their repeat
                # with the kernel timers off, the sustained loop).  Lines of rounds 1 - 4 lack the field: their total is rebuilt from the parts,
                # INCLUDING the timers-off repeat of the K steps that round 4's bench.py ran and round 4's steps_of() forgot (VERDICT round 4,
                # weak 8: every per-step figure of profiles/r04_* is inflated by (W + S + 2 K) / (W + S + K)).
                n = j.get("steps_executed")
                if n is None:
                    n = j["steps"] + j["warmup"] + j.get("spinup_steps_untimed", 0) + (j["steps"] if "without_kernel_timers" in j else 0) + \
                        (j.get("sustained") or {}).get("steps", 0)
                return (n,
                        {"pairs_per_gpu": j["config"].get("pairs_per_gpu"), "num_points": j["config"].get("num_points")})
    except OSError:
        pass
    return None, None


def traffic_summary(pmc, steps_fetch, steps_write, shape, tag, commit):
    """pmc: {kernel: {counter: {"sum", "dispatches"}}}; steps_*: steps the FETCH_SIZE / WRITE_SIZE pass ran."""
    kernels, tot = {}, 0.0
    for k, v in pmc.items():
        f, w = v.get("FETCH_SIZE", {}), v.get("WRITE_SIZE", {})
        if not f and not w:
            continue
        nf, nw = max(f.get("dispatches", 0), 1), max(w.get("dispatches", 0), 1)
        per_launch = (2 * f.get("sum", 0) / nf + w.get("sum", 0) / nw) * 1024
        lps = (f.get("dispatches", 0) / steps_fetch) if f and steps_fetch else (w.get("dispatches", 0) / steps_write if steps_write else 0.0)
        kernels[k] = {"hbm_bytes_per_launch": per_launch, "launches_per_step": lps, "hbm_bytes_per_step": per_launch * lps,
                      "launches": max(f.get("dispatches", 0), w.get("dispatches", 0)),
                      "FETCH_SIZE_KiB": f.get("sum", 0), "WRITE_SIZE_KiB": w.get("sum", 0)}
        tot += per_launch * lps
    return {"tag": tag, "steps_profiled": {"FETCH_SIZE": steps_fetch, "WRITE_SIZE": steps_write}, "shape": shape, "commit": commit,
            "hbm_bytes_per_step": tot,
            "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])),
            "note": "hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per kernel name; per launch = each pass's bytes / that pass's launches; per step = "
                    "per launch x launches per step of the same pass (its own step count from its own bench line).  The x2 is the gfx950 "
                    "FETCH_SIZE correction for wide coalesced reads (MI355X_MICROARCH.md, HBM: an upper bound for narrow reads; WRITE_SIZE "
                    "uncalibrated).  Separate --pmc passes, warm-up and spin-up steps included on both sides."}


if __name__ == "__main__":
    steps_f, shape = steps_of(os.path.join(out, "bench_pmc_FETCH_SIZE.log"))
    steps_w, shape_w = steps_of(os.path.join(out, "bench_pmc_WRITE_SIZE.log"))
    commit = None
    cand = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_commit")
    if os.path.exists(cand):
        commit = open(cand).read().strip()
    if steps_f or steps_w:
        json.dump(traffic_summary(pmc, steps_f, steps_w, shape or shape_w, tag, commit), open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print("summary files:", sorted(os.listdir(dst)))
