"""Soak run of the randomised assembly test beyond the seeds the suite holds: random sorted grids (inside the plan's limits) and
arbitrary ones, every variant of the plan-based build forced in turn, all ten matrices bitwise against the oracle.
usage: fuzz_builds.py [first_seed] [count]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import icebin_amd
import test_gpu_parity as T
orc = T.orc

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    for force_sorted in (True, False):
        g, em = T._random_grid(seed, force_sorted=force_sorted)
        mm, rg = icebin_amd.from_synthetic(g), orc.Regridder(g)
        rm = mm.regrid_matrices("greenland", em)
        icebin_amd.set_tuning("assemble_stream_count", seed % 2)
        icebin_amd.set_tuning("assemble_range_shape", seed % 4)
        icebin_amd.set_tuning("assemble_static_count", (seed // 2) % 2)
        # the streamed build (what grids of 2^20 cells and more take) on every third seed, its row kernels and tile walk varied
        icebin_amd.set_tuning("assemble_stream", 1 if seed % 3 == 0 else -2 ** 31)
        icebin_amd.set_tuning("assemble_stream_rowsl", (seed // 3) % 2)
        icebin_amd.set_tuning("assemble_stream_rowsl_r", (1, 4, 16, 7)[(seed // 6) % 4])
        icebin_amd.set_tuning("assemble_stream_rows4", (seed // 12) % 2)
        icebin_amd.set_tuning("assemble_stream_emit_blocks", (0, 1, 3, 8192)[(seed // 3) % 4])
        icebin_amd.set_tuning("assemble_stream_emit_cpt", (1, 2, 4)[(seed // 9) % 3])
        nfast = 0
        for name in T.ALL:
            for scale, correctA in ((True, True), (False, False), (True, False)):
                try:
                    w = rm.matrix_d(name, scale=scale, correctA=correctA)
                    nfast += int(w.built_fast())
                    T.assert_same_weighted(w, rg.matrix_d(name, em, scale=scale, correctA=correctA), "%s seed=%d" % (name, seed))
                except AssertionError as e:
                    bad += 1
                    print("MISMATCH seed", seed, "sorted", force_sorted, name, scale, correctA, str(e)[:200], flush=True)
        print("seed %d sorted=%s nX=%d fast builds %d/30 ok" % (seed, force_sorted, len(g["ex_area"]), nfast), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
