"""profiles/traffic.json from the two PMC passes (variants/pmc_traffic.py outputs):
    make_traffic.py <fetch.json> <write.json> <out.json>
FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 B, variants/pmc_calib.py), WRITE_SIZE as is."""
import json, sys
f, w, out = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3]
names = {"k_forward": "k_forward_fallback", "k_backward": "k_backward_fallback"}
res = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --no-cpu-baseline "
                "--steps 3 --warmup 1`; counters are KiB per dispatch; FETCH_SIZE doubled (gfx950 tallies 128-B read "
                "requests at 64 B: a 512 MiB copy reads 262162 KiB, variants/pmc_calib.py), WRITE_SIZE as is (the same "
                "copy writes 524288 KiB).  Narrow gathers are not separately calibrated.  k_forward_fallback / "
                "k_backward_fallback: all launches of k_forward / k_backward of the run averaged (one of the k_forward "
                "launches is bench.py's counter pass over all rays)."}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    fs = f.get(k, {}).get("FETCH_SIZE", {}).get("avg", 0.0)
    ws = w.get(k, {}).get("WRITE_SIZE", {}).get("avg", 0.0)
    n = f.get(k, {}).get("FETCH_SIZE", {}).get("launches", 0)
    res[names.get(k, k)] = dict(fetch_size_kib=round(fs, 1), write_size_kib=round(ws, 1), launches=n,
                                hbm_bytes_per_launch=int((2 * fs + ws) * 1024))
json.dump(res, open(out, "w"), indent=1)
tot = sum(v["hbm_bytes_per_launch"] for k, v in res.items() if k != "_note")
print({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in res.items() if k != "_note"}, "total MB", round(tot / 1e6, 1))
