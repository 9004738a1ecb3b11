"""profiles/rNN_vit224_gemm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-cpu --steps 3 --warmup 2`.

    python tools/make_traffic.py <fetch.db> <write.db> <out.json>
bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 averaged over every bf16 GEMM dispatch (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies
the 128-B requests of a wide stream at 64 B -> doubled; both counters are in KiB)."""
import json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    return cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? and "
                       "(kernel_name like '%gemm_blk%_kernel%' or kernel_name like '%gemm_bf16_big_kernel%') group by kernel_name", (counter,)).fetchall()


def main(fetch_db, write_db, out):
    import bench
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    nf, nw = sum(r[2] for r in f), sum(r[2] for r in w)
    fetch = sum(r[1] * r[2] for r in f) / nf
    write = sum(r[1] * r[2] for r in w) / nw
    res = {'bytes_per_launch': (2 * fetch + write) * 1024, 'fetch_size_kib_raw': fetch, 'write_size_kib_raw': write, 'launches': nf,
           'per_kernel': {'FETCH_SIZE_KiB': {r[0][:60]: [r[1], r[2]] for r in f}, 'WRITE_SIZE_KiB': {r[0][:60]: [r[1], r[2]] for r in w}},
           'gemm_source_digest': bench.gemm_source_digest(),
           'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --no-cpu --steps 3 --warmup 2`, mean over all bf16 GEMM '
                   'dispatches; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B wide-stream '
                   'requests at 64 B). Fetch counts L2 misses to the fabric (Infinity-Cache hits included), not HBM only.'}
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res)[:600])


if __name__ == '__main__':
    main(*sys.argv[1:4])
