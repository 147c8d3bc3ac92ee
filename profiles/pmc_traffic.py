"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (one counter per pass, as the MI355X guide prescribes):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile
    python profiles/pmc_traffic.py <out_f/...db> <out_w/...db> profiles/r02_pmc_traffic.json [dominant_kernel other_kernel ...]

(kernel name substrings; default: the round-2 split-precision kernels, dominant one first)

gfx950 correction (calibrated on a 1 GiB device copy, profiles/pmc_probe.py): read bytes = FETCH_SIZE[KB] * 1024 * 2,
write bytes = WRITE_SIZE[KB] * 1024."""
import json
import sqlite3
import sys

KERNELS = ("conv3x3_halo_split", "wgrad3x3_split_kernel", "igemm_fwd_split_kernel", "igemm_wgrad_split_kernel")   # [0] matches both tile variants


def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c.lower()][0]
    out = {}
    for k in KERNELS:
        rows = db.execute("select %s from counters_collection where kernel_name like ? and counter_name = ?" % vcol,
                          ("%" + k + "%", counter)).fetchall()
        vals = [float(r[0]) for r in rows]
        out[k] = (len(vals), sum(vals) / max(len(vals), 1))
    return out


def main():
    global KERNELS
    if len(sys.argv) > 4:
        KERNELS = tuple(sys.argv[4:])
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in KERNELS:
        rd, wr = f[k][1] * 1024 * 2, w[k][1] * 1024
        res[k] = dict(launches_profiled=f[k][0], fetch_size_kb_avg=f[k][1], write_size_kb_avg=w[k][1], read_bytes_per_launch=rd,
                      write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
    d = res[KERNELS[0]]
    out = dict(kernel=KERNELS[0], **d,
               correction="gfx950: read bytes = FETCH_SIZE*1024*2 (calibrated on a 1 GiB device copy in profiles/pmc_probe.py: "
                          "FETCH_SIZE=524304 KB for 2^30 B read, WRITE_SIZE=1048576 KB for 2^30 B written); write bytes = WRITE_SIZE*1024",
               command="rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                       "--no-profile (two separate passes); averages over every launch of the kernel in the run (all layers, fwd + bwd-data)",
               other_kernels={k: res[k] for k in KERNELS[1:]})
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
