// Per-CU store throughput by access pattern (round 5 probe): one 256-thread workgroup per CU writes `bytes_per_wg` of bf16-tile-like output.
//   pattern 0: instruction = 16 rows x 64 B   (the register-direct GEMM epilogue: lane -> (row = lane & 15, 16-byte chunk lane >> 4))
//   pattern 1: instruction =  8 rows x 128 B  (the LDS-staged epilogue: full lines)
//   pattern 2: instruction =  4 rows x 256 B
//   pattern 3: instruction =  1 KiB contiguous
// Row stride 1536 B (a 768-wide bf16 matrix).  nt = non-temporal stores.  Build: hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int PAT, bool NT>
__global__ __launch_bounds__(256) void k(unsigned char* out, size_t wg_stride, int iters, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* base = out + (size_t)blockIdx.x * wg_stride;
    const int LD = 1536;
    int rows_per_inst, row, col;
    if (PAT == 0) { rows_per_inst = 16; row = lane & 15; col = (lane >> 4) * 16; }
    else if (PAT == 1) { rows_per_inst = 8; row = lane >> 3; col = (lane & 7) * 16; }
    else if (PAT == 2) { rows_per_inst = 4; row = lane >> 4; col = (lane & 15) * 16; }
    else { rows_per_inst = 1; row = 0; col = lane * 16; }
    const u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    const long long t0 = __builtin_readcyclecounter();
    // each wave writes `iters` instructions; consecutive instructions walk down the rows of a 128-row x 256-B sub-tile like an epilogue
    for (int i = 0; i < iters; ++i) {
        int r, c;
        if (PAT == 0) { r = (i >> 2) * 16 + row; c = (i & 3) * 64 + col; }             // 4 instructions complete 16 rows x 256 B
        else if (PAT == 1) { r = (i >> 1) * 8 + row; c = (i & 1) * 128 + col; }
        else if (PAT == 2) { r = i * 4 + row; c = col; }
        else { r = i * 4; c = col; }                                                     // (1 KiB contiguous = 4 rows of 256 B back to back)
        unsigned char* p = (PAT == 3) ? base + (size_t)wave * 32768 * 8 + (size_t)i * 1024 + c
                                      : base + ((size_t)wave * 128 * 8 + (r % (128 * 8))) * LD + c;
        if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
}
template <int PAT, bool NT> void run(unsigned char* d, unsigned long long* dc, int nwg, int iters, const char* name) {
    const size_t stride = (size_t)8 << 20;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL((k<PAT, NT>), dim3(nwg), dim3(256), 0, 0, d, stride, iters, dc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(nwg);
        hipMemcpy(h.data(), dc, nwg * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto x : h) avg += x; avg /= nwg;
        const double bytes = 4.0 * iters * 1024;
        if (rep == 2) printf("%-34s wgs %3d  %6.0f KB/wg  %8.0f cyc/wg  %5.1f B/cyc/CU  launch %7.1f us  %6.2f TB/s\n", name, nwg, bytes / 1024, avg, bytes / avg,
                             ms * 1e3, bytes * nwg / ms / 1e9);
    }
}
int main() {
    unsigned char* d; unsigned long long* dc;
    hipMalloc(&d, (size_t)256 * (8 << 20)); hipMalloc(&dc, 256 * 8);
    for (int nwg : {1, 32, 256}) for (int iters : {32, 256}) {
        run<0, true>(d, dc, nwg, iters, "16 rows x 64 B, nt");
        run<1, true>(d, dc, nwg, iters, "8 rows x 128 B, nt");
        run<2, true>(d, dc, nwg, iters, "4 rows x 256 B, nt");
        run<3, true>(d, dc, nwg, iters, "1 KiB contiguous, nt");
        run<0, false>(d, dc, nwg, iters, "16 rows x 64 B");
        run<1, false>(d, dc, nwg, iters, "8 rows x 128 B");
        run<2, false>(d, dc, nwg, iters, "4 rows x 256 B");
        run<3, false>(d, dc, nwg, iters, "1 KiB contiguous");
    }
    return 0;
}
