"""Where the listing step of the sharded per-file loop (catfish/catfish:49-50) spends its time when N ranks list ONE directory at once:
N processes, a barrier, then each times  DirListing open (readdir + order + digest) / sizes of its n/N block / names of its block,
and -- for comparison -- sorted(os.listdir()).  No GPU involved; run it on the box whose file system is in question.
usage: python tools/exp_listing.py [ranks=4] [files_per_rank=12500] [directory (default: a fresh one under $TMPDIR)]"""
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, directory, barrier, out):
    from catfish_amd import sharding
    sharding.DirListing(os.path.dirname(directory)).close()          # library loaded before the clock starts
    res = {}
    for what in ("native", "python", "native"):
        barrier.wait()
        t0 = time.perf_counter()
        if what == "native":
            lst = sharding.DirListing(directory)
            t1 = time.perf_counter()
            lo, hi = rank * len(lst) // world, (rank + 1) * len(lst) // world
            lst.sizes(lo, hi)
            t2 = time.perf_counter()
            lst.names(lo, hi)
            t3 = time.perf_counter()
            res[what] = (t1 - t0, t2 - t1, t3 - t2)
            lst.close()
        else:
            names = os.listdir(directory)
            t1 = time.perf_counter()
            names.sort()
            t2 = time.perf_counter()
            res[what] = (t1 - t0, t2 - t1)
    out.put((rank, res))


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    per_rank = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
    made = len(sys.argv) <= 3
    directory = tempfile.mkdtemp(prefix="catfish_listing_") if made else sys.argv[3]
    try:
        if made:
            for i in range(world * per_rank):
                with open(os.path.join(directory, "read_%06d.npy" % i), "wb") as fh:
                    fh.write(b"\0" * 64)
        ctx = mp.get_context("spawn")
        barrier, out = ctx.Barrier(world), ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, directory, barrier, out)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted(out.get() for _ in procs)
        for p in procs:
            p.join()
        print("%d ranks x %d files in %s (%s)" % (world, per_rank, directory, os.popen("df -T %s | tail -1" % directory).read().split()[1]))
        for rank, res in got:
            n = res["native"]
            print("rank %d: native open %.1f ms, sizes of block %.1f, names of block %.1f | python listdir %.1f, sort %.1f" % (
                rank, n[0] * 1e3, n[1] * 1e3, n[2] * 1e3, res["python"][0] * 1e3, res["python"][1] * 1e3))
    finally:
        if made:
            shutil.rmtree(directory, ignore_errors=True)


if __name__ == "__main__":
    main()
