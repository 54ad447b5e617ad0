"""usage: collect_profiles.py [round tag, default r03].  Copy the summaries tools/profile_hpsi.sh left under gpurun_out/prof_<round>/<workload>/ into profiles/<round>_<workload>_* and
rebuild profiles/pmc_traffic.json (bench.py reads `roofline.traffic` and its source from there).
FETCH_SIZE counts KiB and reports half of the bytes on gfx950 (x2, MI355X_MICROARCH.md, confirmed by the vec_axpy_k
calibration dispatch in every file: 2 GiB read -> 1048588); WRITE_SIZE counts KiB."""
import json, os, re, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(root, "gpurun_out", "prof_" + rnd)
out = {}
for w in sorted(os.listdir(src)):
    d = os.path.join(src, w)
    if not os.path.exists(os.path.join(d, "pmc_fetch_write.txt")):
        continue
    for f, t in (("kernel_stats.csv", "kernel_stats.csv"), ("pmc_fetch_write.txt", "pmc_fetch_write.txt"),
                 ("bench_under_rocprof.json", "bench_under_rocprof.json")):
        shutil.copy(os.path.join(d, f), os.path.join(root, "profiles", "%s_%s_%s" % (rnd, w, t)))
    tot, calib, which = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}, {}, None
    for line in open(os.path.join(d, "pmc_fetch_write.txt")):
        m = re.match(r"# (\w+) summed", line)
        if m:
            which = m.group(1)
            continue
        m = re.match(r"(.*?)\s+dispatches\s+(\d+)\s+sum\s+(\S+)", line)
        if not m:
            continue
        name, val = m.group(1), float(m.group(3))
        if "vec_axpy_k" in name:
            calib[which] = val
        elif "b2x::" in name:
            tot[which] += val
    out[w] = {"fetch_bytes_per_hpsi": tot["FETCH_SIZE"] * 2 * 1024, "write_bytes_per_hpsi": tot["WRITE_SIZE"] * 1024,
              "source": "profiles/%s_%s_pmc_fetch_write.txt (all b2x kernels of one H.psi; FETCH_SIZE in KiB x2 (gfx950 "
                        "correction), WRITE_SIZE in KiB; calibration vec_axpy_k in the same file: 2 GiB read -> %d, 1 GiB "
                        "written -> %d)" % (rnd, w, calib.get("FETCH_SIZE", 0), calib.get("WRITE_SIZE", 0))}
    print("%-16s FETCH %.4g GB  WRITE %.4g GB" % (w, out[w]["fetch_bytes_per_hpsi"] / 1e9, out[w]["write_bytes_per_hpsi"] / 1e9))
tf = os.path.join(root, "profiles", "pmc_traffic.json")
old = json.load(open(tf)) if os.path.exists(tf) else {}
old.update(out)  # (workloads profiled in earlier rounds keep their entry and its source)
json.dump(old, open(tf, "w"), indent=1, sort_keys=True)
