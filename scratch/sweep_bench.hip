// Microbenchmark 3: the column-sweep form of an E-row apply (spmm_sweep_kernel) on a synthetic 1 km EvI-shaped structure.
// Groups of GW consecutive columns (one GCM cell), every column with two entries (classes k, k+1; the class changes
// every RUN columns), X[nf=64][ldx] field-major.  Checks group 0 against a host loop and reports us / TB/s.
// usage: sweep_bench ncol GW [ldx]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#include "../icebin_amd/csrc/sweep_kernel.inl"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int ncol = argc > 1 ? atoi(argv[1]) : 1907329;
    const int GW = argc > 2 ? atoi(argv[2]) : 3818;
    const long ldx = argc > 3 ? atol(argv[3]) : ncol;
    const int RUN = 17, NSL = 8, nf = 64, CB = ibh::SWEEP_CB;
    const int nbuf = ncol > 1000000 ? 3 : 14;
    const int ngroup = (ncol + GW - 1) / GW;
    const size_t xn = (size_t)nf * ldx;
    double *d_X; CK(hipMalloc(&d_X, 8 * xn * nbuf));
    std::vector<double> hx(xn);
    for (size_t i = 0; i < xn; ++i) hx[i] = std::sin(0.001 * (double)(i % 100003)) + 1.5;
    for (int b = 0; b < nbuf; ++b) CK(hipMemcpy(d_X + xn * b, hx.data(), 8 * xn, hipMemcpyHostToDevice));
    {
        hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
        int o0 = 0, o1 = 0, o2 = 0;
        const size_t lds = ibh::sweep_lds_bytes(NSL);
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o0, ibh::spmm_sweep_kernel<true, false, 0>, 256, lds));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o1, ibh::spmm_sweep_kernel<true, true, 0>, 256, lds));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, ibh::spmm_sweep_kernel<true, true, 1>, 256, lds));
        printf("device: %s CUs %d clock %d kHz LDS/CU %zu LDS/block %zu regs/block %d; lds per WG %zu; occupancy (WGs/CU) MODE0 %d MODE1 %d MODE2 %d\n", pr.name,
               pr.multiProcessorCount, pr.clockRate, pr.maxSharedMemoryPerMultiProcessor, pr.sharedMemPerBlock, pr.regsPerBlock, lds, o0, o1, o2);
    }
    for (int TB : {2, 8, 16}) for (int IDENT : {0, 1}) for (int MODE : {0, 1}) {
        // packed structure: one item per column, 64 items a block, TB blocks a task, NSL slots per task
        std::vector<int> it_col, task_p0, task_ns;
        std::vector<unsigned> meta; std::vector<double> val0, val1;
        const int nblk = (ncol + CB - 1) / CB, ntask = (nblk + TB - 1) / TB;
        for (int c = 0; c < nblk * CB; ++c) {
            if (c >= ncol) { it_col.push_back(0); meta.push_back(0u); val0.push_back(0.0); val1.push_back(0.0); continue; }
            const int s = ((c % GW) / RUN) % (NSL - 1);
            const double t = 0.25 + 0.5 * ((c * 7) % 13) / 13.0;
            it_col.push_back(c);
            meta.push_back((unsigned)s | ((unsigned)(s + 1) << 8) | ibh::SWEEP_HAS0 | ibh::SWEEP_HAS1);
            val0.push_back((1 - t) / GW); val1.push_back(t / GW);
        }
        for (int t = 0; t < ntask; ++t) { task_p0.push_back(t * NSL); task_ns.push_back(NSL); }
        const int prow = ntask * NSL;
        int *d_tp0, *d_tns, *d_brg; unsigned *d_meta; double *d_val0, *d_val1, *d_P;
        CK(hipMalloc(&d_tp0, 4 * ntask)); CK(hipMalloc(&d_tns, 4 * ntask));
        CK(hipMalloc(&d_brg, 4 * it_col.size()));
        CK(hipMalloc(&d_meta, 4 * meta.size())); CK(hipMalloc(&d_val0, 8 * val0.size())); CK(hipMalloc(&d_val1, 8 * val1.size()));
        CK(hipMemcpy(d_tp0, task_p0.data(), 4 * ntask, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_tns, task_ns.data(), 4 * ntask, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_brg, it_col.data(), 4 * it_col.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_meta, meta.data(), 4 * meta.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_val0, val0.data(), 8 * val0.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_val1, val1.data(), 8 * val1.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_P, 8 * (size_t)prow * 64));

        ibh::SweepView sv{d_tp0, d_tns, d_brg, d_meta, d_val0, d_val1, TB, nblk, ncol};
        ibh::SweepBatch sb{};
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t lds = ibh::sweep_lds_bytes(NSL);
        auto launch = [&](int i) {
            sb.x[0] = d_X + xn * (i % nbuf); sb.p[0] = d_P;
#define L_(I, M) hipLaunchKernelGGL((ibh::spmm_sweep_kernel<true, I, M>), dim3(ntask, 1, 1), dim3(64 * ibh::SWEEP_NW), lds, 0, sv, sb, ldx, nf, NSL, 64l, 6, 1)
            if (IDENT) { if (MODE == 0) L_(true, 0); else L_(true, 1); }
            else { if (MODE == 0) L_(false, 0); else L_(false, 1); }
        };
        for (int i = 0; i < 5; ++i) launch(i);
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        const int steps = ncol > 1000000 ? 20 : 200;
        CK(hipEventRecord(e0));
        for (int i = 0; i < steps; ++i) launch(i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / steps;
        std::vector<double> hp((size_t)prow * 64);
        launch(0); CK(hipDeviceSynchronize());
        CK(hipMemcpy(hp.data(), d_P, 8 * hp.size(), hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int f = 0; f < nf; f += 21) {
            std::vector<double> ref(NSL, 0.0);
            for (int e = 0; e < TB * CB && e < ncol; ++e) {
                const double xv = hx[(size_t)f * ldx + it_col[e]];
                ref[meta[e] & 255] += val0[e] * xv; ref[(meta[e] >> 8) & 255] += val1[e] * xv;
            }
            for (int s2 = 0; s2 < NSL; ++s2) maxerr = std::max(maxerr, std::fabs(ref[s2] - hp[(size_t)s2 * 64 + f]));
        }
        printf("ncol %d ldx %ld TB %3d IDENT %d MODE %d: ntask %6d nblk %7d  %8.2f us  %6.2f TB/s  (group-0 check max err %.2e)\n", ncol, ldx, TB, IDENT, MODE,
               ntask, nblk, us, (double)nf * ncol * 8 / us / 1e6, maxerr);
        fflush(stdout);
        hipFree(d_tp0); hipFree(d_tns); hipFree(d_brg); hipFree(d_meta); hipFree(d_val0); hipFree(d_val1); hipFree(d_P);
    }
    return 0;
}
