"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv) into per-launch HBM traffic per kernel.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reports half the
bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and
float atomics.  Narrower access patterns are uncalibrated: the figures are an upper-bound style estimate.
"""
import csv, glob, json, sys
from collections import defaultdict

def load(path, counter):
    f = glob.glob(path + "/*/*counter_collection.csv")[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
def add(fetch, write, prefix="", only=None):
    for k in sorted(set(fetch) | set(write)):
        if only and only not in k:
            continue
        f = fetch.get(k, [0.0]); w = write.get(k, [0.0])
        fb = 1024.0 * sorted(f)[len(f) // 2]; wb = 1024.0 * sorted(w)[len(w) // 2]
        out[prefix + k] = {"launches": len(f), "fetch_bytes_raw": fb, "write_bytes": wb, "hbm_bytes_corrected": 2 * fb + wb}
add(fetch, write)
if len(sys.argv) > 5:  # a second pair of passes over `roofline_kernels.py cross`: the cross-attention launches
    add(load(sys.argv[4], "FETCH_SIZE"), load(sys.argv[5], "WRITE_SIZE"), prefix="sdpa_fwd_cross: ", only="sdpa_fwd")
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{v['hbm_bytes_corrected']/1e6:10.2f} MB/launch (fetch raw {v['fetch_bytes_raw']/1e6:8.2f}, write {v['write_bytes']/1e6:8.2f})  {k[:100]}")
