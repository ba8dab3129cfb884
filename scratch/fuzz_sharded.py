"""Soak run of the SHARDED assembly (ibh_regrid_matrices_matrix_d_sharded) on the random sorted grids of the assembly tests: worlds of
2, 3 and 4 ranks sharing the box's card over the host-staged gloo transport of tests/test_distributed_gloo.py, every matrix the
shared build serves, identity / pre-populated sets and the coupler's four calls, bitwise the single-rank build.
usage: fuzz_sharded.py [first_seed] [count] [seeds per spawn]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch.multiprocessing as mp
import test_distributed_gloo as D

if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    bad = 0
    ctx = mp.get_context("spawn")
    for s0 in range(first, first + count, per):
        world = 2 + (s0 // per) % 3
        configs = tuple("rand:%d" % s for s in range(s0, min(s0 + per, first + count)))
        q = ctx.Queue()
        port = D._free_port()
        procs = [ctx.Process(target=D._worker_asm_sharded, args=(r, world, port, q, configs, 64)) for r in range(world)]
        for p in procs: p.start()
        for p in procs: p.join(600)
        codes = [p.exitcode for p in procs]
        got = sorted(q.get(timeout=5) for _ in range(world)) if all(c == 0 for c in codes) else []
        okay = bool(got) and all(ok for _, ok, _, _ in got)
        if not okay:
            bad += 1
        print("seeds %d..%d world %d: %s  exit codes %s  %s" % (s0, s0 + len(configs) - 1, world, "ok" if okay else "FAILED", codes,
              ("declined (built redundantly): %s" % sorted(set(" ".join(x.split()[:2]) for x in got[0][2]))) if okay and got[0][2] else "" if okay else [n for _, _, n, _ in got]), flush=True)
    print("batches failed:", bad)
    sys.exit(1 if bad else 0)
