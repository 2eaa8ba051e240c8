// psgdk.hip -- C-ABI implementation (see include/psgdk.h).  One translation unit; kernels live in the .hiph files.
#include "host_util.hiph"
#include "kernels_ew.hiph"

extern "C" {

int psgdk_version(void) { return PSGDK_VERSION; }

const char* psgdk_strerror(int status) {
    switch (status) {
        case PSGDK_OK: return "ok";
        case PSGDK_ERR_INVALID: return "invalid argument";
        case PSGDK_ERR_UNSUPPORTED: return "unsupported (not built yet)";
        case PSGDK_ERR_HIP: return "HIP runtime error (see psgdk_last_hip_error)";
        case PSGDK_ERR_STATE: return "call order violated";
        default: return "unknown status";
    }
}

int psgdk_last_hip_error(void) { return g_last_hip_error; }

int psgdk_fill_normal(void* out, int dtype, int64_t n, uint64_t seed, uint64_t offset, uint32_t stream_id,
                      void* stream) {
    if (!out || n < 0 || (dtype != PSGDK_BF16 && dtype != PSGDK_F32)) return PSGDK_ERR_INVALID;
    if (n == 0) return PSGDK_OK;
    const int block = 256;
    const int grid = (int)std::min<int64_t>((n + block - 1) / block, 256 * 8);
    hipLaunchKernelGGL(fill_normal_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, out, dtype, n, seed, offset,
                       stream_id);
    HIPCHK(hipGetLastError());
    return PSGDK_OK;
}

int psgdk_test_gemm_nt(const void* A, const void* B, void* C, void* Ct, int dtype, int M, int N, int K, int lda,
                       int ldb, int ldc, int ldct, int symmetric, void* stream) {
    if (!A || !B || (!C && !Ct) || (M % 64) || (N % 64) || (K % 64) || M <= 0 || N <= 0 || K <= 0)
        return PSGDK_ERR_INVALID;
    if (dtype != PSGDK_BF16 && dtype != PSGDK_F32) return PSGDK_ERR_INVALID;
    GemmProblem P{};
    P.A = A; P.B = B; P.C = C; P.Ct = Ct;
    P.M = M; P.N = N; P.K = K; P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.ldct = ldct;
    P.alpha = 1.0f; P.flags = symmetric ? GF_SYM : 0;
    if (symmetric) { if (M != N || !C) return PSGDK_ERR_INVALID; P.Ct = C; P.ldct = ldc; }
    TileTableBuilder tb;
    tb.add_problem(0, P);
    std::vector<GemmTile> tiles = tb.finish();
    GemmProblem* dP = nullptr; GemmTile* dT = nullptr;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMalloc(&dP, sizeof(P)));
    HIPCHK(hipMalloc(&dT, tiles.size() * sizeof(GemmTile)));
    HIPCHK(hipMemcpyAsync(dP, &P, sizeof(P), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dT, tiles.data(), tiles.size() * sizeof(GemmTile), hipMemcpyHostToDevice, s));
    if (dtype == PSGDK_BF16)
        hipLaunchKernelGGL(gemm_nt_kernel<bf16_t>, dim3((unsigned)tiles.size()), dim3(256), 0, s, dP, dT);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<float>, dim3((unsigned)tiles.size()), dim3(256), 0, s, dP, dT);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(dP));
    HIPCHK(hipFree(dT));
    return PSGDK_OK;
}

}  // extern "C"
