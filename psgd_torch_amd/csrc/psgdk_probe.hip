// psgdk_probe.hip -- libpsgdk_probe.so: in-process ceilings of THIS chip (bench.py's roofline.peak_measured; SURVEY 8d "re-verify on the
// box").  A measurement tool, NOT part of the product library: libpsgdk.so carries no probe kernels.  Declared in include/psgdk_test.h.
#include "kernels_probe.hiph"
#include "../../include/psgdk_test.h"
#include <algorithm>
#include <vector>

#define HIPCHK(x) do { if ((x) != hipSuccess) return PSGDK_ERR_HIP; } while (0)

extern "C" {

int psgdk_test_clock(float* shader_mhz, void* stream) {
    if (!shader_mhz) return PSGDK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)cus * 2;
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)grid * 2 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h((size_t)grid * 2);
    int rc = PSGDK_OK;
    for (int rep = 0; rep < 2; ++rep)          // (the first launch brings the clocks up)
        hipLaunchKernelGGL(clock_probe_kernel, dim3(grid), dim3(256), 0, st, d, 40000);
    if (hipMemcpyAsync(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) rc = PSGDK_ERR_HIP;
    (void)hipFree(d);
    if (rc) return rc;
    std::vector<double> mhz;
    for (unsigned b = 0; b < grid; ++b) if (h[2 * b + 1] > 0) mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
    if (mhz.empty()) return PSGDK_ERR_HIP;
    std::nth_element(mhz.begin(), mhz.begin() + mhz.size() / 2, mhz.end());
    *shader_mhz = (float)mhz[mhz.size() / 2];
    return PSGDK_OK;
}

int psgdk_test_peaks(float* out4, void* scratch, size_t scratch_bytes, void* stream) {
    if (!out4 || !scratch || scratch_bytes < (size_t)64 << 20) return PSGDK_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int rc = PSGDK_OK;
    auto timed = [&](auto&& launch, int reps) -> float {          // best of `reps` launches, ms
        float best = 1e30f;
        launch();                                                  // warm-up (clocks, code)
        for (int r = 0; r < reps; ++r) {
            (void)hipEventRecord(e0, st);
            launch();
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = PSGDK_ERR_HIP; return best; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        return best;
    };
    const int iters = 4000;
    const unsigned grid = (unsigned)cus * 2;                       // two waves per SIMD
    const double flops = (double)grid * 4 * iters * 16 * 16384.0;  // per wave and round: 16 MFMAs of 16 x 16 x 32 (2^18 FLOP); the 32 x 32 x 16 round is 16 MFMAs of twice that
    float* sink = (float*)scratch;
    const float ms16 = timed([&] { hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(grid), dim3(256), 0, st, sink, iters); }, 3);
    const float ms32 = timed([&] { hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(grid), dim3(256), 0, st, sink, iters); }, 3);
    out4[0] = (float)(flops / (ms16 * 1e-3) / 1e12);
    out4[1] = (float)(2.0 * flops / (ms32 * 1e-3) / 1e12);
    const size_t half = (scratch_bytes / 2) & ~(size_t)255, n = half / 16;
    const u32x4_t* src = (const u32x4_t*)scratch;
    u32x4_t* dst = (u32x4_t*)((unsigned char*)scratch + half);
    // two launch shapes each (a resident grid with four chunks in flight per thread; one chunk per thread): the better one counts
    const unsigned cgrid = (unsigned)cus * 8, fgrid = (unsigned)((n + 255) / 256);
    const float msc = std::min(timed([&] { hipLaunchKernelGGL(hbm_copy_kernel, dim3(cgrid), dim3(256), 0, st, src, dst, n, 0); }, 3),
                               timed([&] { hipLaunchKernelGGL(hbm_copy_kernel, dim3(fgrid), dim3(256), 0, st, src, dst, n, 0); }, 3));
    const float msr = std::min(timed([&] { hipLaunchKernelGGL(hbm_copy_kernel, dim3(cgrid), dim3(256), 0, st, src, dst, n, 1); }, 3),
                               timed([&] { hipLaunchKernelGGL(hbm_copy_kernel, dim3(fgrid), dim3(256), 0, st, src, dst, n, 1); }, 3));
    out4[2] = (float)(2.0 * (double)half / (msc * 1e-3) / 1e9);    // read + write
    out4[3] = (float)((double)half / (msr * 1e-3) / 1e9);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (hipGetLastError() != hipSuccess) rc = PSGDK_ERR_HIP;
    return rc;
}


}  // extern "C"
