"""Folds the rocprofv3 outputs of variants/pmc_profile.sh into profiles/<tag>_counters.json (+ profiles/counters.json,
the file bench.py reads) and copies the kernel-trace summary to profiles/<tag>_kernel_stats.csv.
    make_counters.py <dir> <tag> <the bench.py command line the counters were collected with ...>
Per kernel: launches, avg duration (kernel trace), FETCH_SIZE / WRITE_SIZE (KiB per dispatch), hbm_bytes =
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 correction of MI355X_MICROARCH.md, checked here against a 512 MiB copy),
write_bytes, and the SQ counters (chip-wide sums per dispatch)."""
import csv, glob, json, os, re, shutil, sys

root, tag, cmd = sys.argv[1], sys.argv[2], sys.argv[3:]
RENAME = {"k_forward": "k_forward_fallback", "k_backward": "k_backward_fallback", "k_sweep_wave": "k_sweep"}   # (bench.py's span names)


def short(name):
    s = re.sub(r"\(.*", "", name).replace("void ", "")
    s = re.sub(r"<.*", "", s).replace("asurf::", "")
    return s


def fold(sub):
    acc = {}
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (tot / n, n) for k, (tot, n) in acc.items()}


fetch, write, sq, atomic = fold("fetch"), fold("write"), fold("sq"), fold("atomic")
cf, cw = fold("calib_fetch"), fold("calib_write")
calib = {}
for (k, c), (avg, n) in list(cf.items()) + list(cw.items()):
    if "elementwise" in k or "copy" in k.lower() or "fill" in k.lower():
        calib[f"{k[:40]}:{c}"] = round(avg, 1)

# kernel-trace stats
dur = {}
stats = sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True))
if stats:
    shutil.copy(stats[-1], os.path.join("profiles", f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[-1])):
        dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e6, int(r["Calls"]))

# the configuration: parsed from the command line exactly as bench.py does
sys.argv = cmd
sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
a = bench.parse()
cfg = bench.config_key(a)

kernels = {}
names = sorted({k for k, _ in list(fetch) + list(write) + list(sq) if k.startswith("k_")})
for k in names:
    fs, nf = fetch.get((k, "FETCH_SIZE"), (0.0, 0))
    ws, _ = write.get((k, "WRITE_SIZE"), (0.0, 0))
    e = dict(launches=nf, fetch_size_kib=round(fs, 1), write_size_kib=round(ws, 1),
             hbm_bytes=int((2 * fs + ws) * 1024), write_bytes=int(ws * 1024))
    for c, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_SALU", "salu_insts"), ("SQ_INSTS_LDS", "lds_insts"),
                   ("SQ_INSTS_VMEM", "vmem_insts"), ("SQ_WAVES", "waves"), ("SQ_WAVE_CYCLES", "wave_cycles"),
                   ("SQ_WAIT_ANY", "wait_any"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu")):
        if (k, c) in sq:
            e[key] = round(sq[(k, c)][0], 1)
    if (k, "TCC_ATOMIC_sum") in atomic:
        e["atomic_requests"] = round(atomic[(k, "TCC_ATOMIC_sum")][0], 1)
    if k in dur:
        e["trace_avg_ms"], e["trace_calls"] = round(dur[k][0], 5), dur[k][1]
    kernels[RENAME.get(k, k)] = e

out = dict(
    config=cfg, command="python3 " + " ".join(cmd),
    note="rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ counters: separate runs, no trace domain beside them) and a "
         "--kernel-trace --stats run of the same command (variants/pmc_profile.sh).  Counter values are averages per "
         "dispatch over all launches of the run; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: gfx950 tallies "
         "128-byte read requests at 64 B (MI355X_MICROARCH.md, HBM section; `calibration` below: a 512 MiB float copy and "
         "fill under the same passes).  Narrow gathers are not separately calibrated.  k_forward_fallback / "
         "k_backward_fallback: every launch of k_forward / k_backward of the run averaged (one k_forward launch is "
         "bench.py's counter pass over all rays).",
    calibration=calib, kernels=kernels)
json.dump(out, open(os.path.join("profiles", f"{tag}_counters.json"), "w"), indent=1)
# profiles/counters.json: the headline's set at the top level, the sets of other workloads under "sets"
main_p = os.path.join("profiles", "counters.json")
try:
    main = json.load(open(main_p))
except Exception:  # noqa: BLE001
    main = None
is_headline = (cfg["reso"] == 256 and cfg["basis_dim"] == 9 and cfg["width"] == 800 and not cfg["shuffle"] and not cfg["heavy"]
               and not cfg["links_epoch"] and not cfg["eikonal"] and cfg["band"] == 3.0)
if is_headline or main is None:
    sets = [] if main is None else [x for x in main.get("sets", []) if x.get("config") != cfg]
    out_main = dict(out, sets=sets)
else:
    sets = [x for x in main.get("sets", []) if x.get("config") != cfg] + [out]
    out_main = dict(main, sets=sets)
json.dump(out_main, open(main_p, "w"), indent=1)
print(json.dumps({k: dict(hbm_MB=round(v["hbm_bytes"] / 1e6, 1), valu_M=round(v.get("valu_insts", 0) / 1e6, 2),
                          ms=v.get("trace_avg_ms")) for k, v in kernels.items()}, indent=1))
