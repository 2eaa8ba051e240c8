// Per-CU rate of the GEMM operand stream by path (round 5 probe): workgroups of 256 or 512 threads read bf16-tile-like rows (8 rows x 128 B per
// wave-instruction, row stride 1536 B) that are L2-resident (each XCD's workgroups walk the same 2 MiB again and again) into LDS,
//   path 0: LDS-DMA only            (global_load_lds_dwordx4: what both GEMM kernels use)
//   path 1: registers + ds_write    (global_load_dwordx4, then ds_write_b128)
//   path 2: alternating pieces, half each
// pieces = 1 KiB wave-instructions; PF pieces are in flight per wave before the first wait.  Build: hipcc --offload-arch=gfx950 -O3 operand_paths.hip -o operand_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(1))) const void* gvoid_t;
typedef __attribute__((address_space(3))) void* lvoid_t;

template <int PATH, int PF>
__global__ void k(const unsigned char* __restrict__ src, int rounds, unsigned long long* cyc, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int LD = 1536;
    // the XCD's shared 2 MiB window: block b runs on XCD b % 8
    const unsigned char* base = src + (size_t)(blockIdx.x & 7) * (2u << 20);
    const int row = lane >> 3, col = (lane & 7) * 16;
    unsigned char* my = lds + wave * PF * 1024;
    unsigned acc = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            // piece (r, wave, i): 8 rows of 128 B somewhere in the window
            const size_t piece = ((size_t)r * nw + wave) * PF + i;
            const unsigned char* p = base + ((piece * 8 + row) % 1360) * LD + ((piece >> 3) % 10) * 128 + col;
            const bool dma = PATH == 0 || (PATH == 2 && (i & 1) == 0);
            if (dma) __builtin_amdgcn_global_load_lds((gvoid_t)p, (lvoid_t)(my + i * 1024), 16, 0, 0);
            else v[i] = *(const u32x4*)p;
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const bool dma = PATH == 0 || (PATH == 2 && (i & 1) == 0);
            if (!dma) *(u32x4*)(my + i * 1024 + lane * 16) = v[i];
        }
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0), lgkmcnt(0)
        if (r == rounds - 1) acc += *(unsigned*)(my + lane * 4);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PATH, int PF> void run(const unsigned char* d, unsigned long long* dc, unsigned* sink, int nwg, int threads, int rounds, const char* name) {
    const int nw = threads / 64;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL((k<PATH, PF>), dim3(nwg), dim3(threads), nw * PF * 1024, 0, d, rounds, dc, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(nwg);
        hipMemcpy(h.data(), dc, nwg * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto x : h) avg += x; avg /= nwg;
        const double bytes = (double)rounds * nw * PF * 1024;
        const int wg_per_cu = nwg > 256 ? nwg / 256 : 1;
        if (rep == 2) printf("%-28s PF %d  wgs %3d x %3d thr  %6.0f KB/wg  %8.0f cyc/wg  %5.1f B/cyc/CU  launch %7.1f us\n", name, PF, nwg, threads, bytes / 1024, avg,
                             bytes * wg_per_cu / avg, ms * 1e3);
    }
}
int main() {
    unsigned char* d; unsigned long long* dc; unsigned* sink;
    hipMalloc(&d, (size_t)8 * (2u << 20) + (4u << 20)); hipMalloc(&dc, 1024 * 8); hipMalloc(&sink, 4);
    hipMemset(d, 1, (size_t)8 * (2u << 20) + (4u << 20));
    for (int nwg : {256, 512}) for (int threads : {256, 512}) {
        if (nwg == 512 && threads == 512) continue;
        const int rounds = 512;
        run<0, 4>(d, dc, sink, nwg, threads, rounds, "LDS-DMA");
        run<1, 4>(d, dc, sink, nwg, threads, rounds, "registers + ds_write");
        run<2, 4>(d, dc, sink, nwg, threads, rounds, "half / half");
        run<0, 8>(d, dc, sink, nwg, threads, rounds, "LDS-DMA");
        run<1, 8>(d, dc, sink, nwg, threads, rounds, "registers + ds_write");
        run<2, 8>(d, dc, sink, nwg, threads, rounds, "half / half");
    }
    return 0;
}
