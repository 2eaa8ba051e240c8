/* The drop-in boundary used from plain C: no Python, no torch -- the library's only dependencies are the HIP runtime and
 * libstdc++.  This is what a binding in any host language does: plan the tensors, allocate the two
 * arenas, then per step accumulate -> update_precond -> precond_grad -> apply_update on raw device pointers.
 *
 * Here: a (96, 64) weight and a (64,) bias, 60 whitening steps on gradients g = diag(s) z whose row scales s run from 0.25 to 4
 * (row covariance diag(s^2), condition number 256): the row covariance of the preconditioned gradient h = P g must come out much
 * closer to a multiple of the identity than that of g itself (psgd.py:394-419 whitens: E[h h^T] -> I), the state must stay finite
 * and the parameters must move.  misc/psgd_kron_verification.py's idea at toy size; the CPU oracle reaches 0.098 vs 0.188 (ratio
 * 0.52) on this metric over the last ten steps, the sampling floor of a 96 x 64 sample being about 0.1.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_abi/abi_demo.c -o tests/c_abi/abi_demo \
 *       -L psgd_torch_amd -l:libpsgdk.so -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN/../../psgd_torch_amd' -Wl,-rpath,/opt/rocm/lib
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "psgdk.h"

#define CK(x) do { int rc_ = (x); if (rc_ != PSGDK_OK) { fprintf(stderr, "%s -> %s (hip %d)\n", #x, psgdk_strerror(rc_), psgdk_last_hip_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> hip error %d\n", #x, (int)e_); return 1; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static double urand(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0; }
static double nrand(void) { double u1 = urand() + 1e-300, u2 = urand(); return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2); }

int main(void) {
    enum { R = 96, Cc = 64, NW = R * Cc, NB = Cc, STEPS = 60 };
    if (psgdk_version() != PSGDK_VERSION) { fprintf(stderr, "unexpected library version\n"); return 1; }
    const int32_t ndim[2] = {2, 1};
    const int64_t dims[3] = {R, Cc, Cc};
    psgdk_plan* plan = NULL;
    CK(psgdk_plan_create(&plan, 2, ndim, dims, INFINITY, INFINITY, PSGDK_F32, 1));        /* max_skew = inf: both dims of W dense */
    size_t sb = 0, wb = 0;
    CK(psgdk_plan_arena_bytes(plan, &sb, &wb));
    void *state = NULL, *work = NULL;
    HK(hipMalloc(&state, sb)); HK(hipMalloc(&work, wb));
    CK(psgdk_plan_bind(plan, state, work));
    CK(psgdk_init_state(plan, 1.0, NULL));
    int nf = 0;
    CK(psgdk_plan_num_factors(plan, 0, &nf));
    if (nf != 2) { fprintf(stderr, "expected two Kronecker factors for the weight, got %d\n", nf); return 1; }

    float *hW = malloc(sizeof(float) * NW), *hB = malloc(sizeof(float) * NB), *hG = malloc(sizeof(float) * NW), *hGb = malloc(sizeof(float) * NB);
    float *hH = malloc(sizeof(float) * NW), *sc = malloc(sizeof(float) * R);
    for (int i = 0; i < NW; ++i) hW[i] = 0.1f * (float)nrand();
    for (int i = 0; i < NB; ++i) hB[i] = 0.f;
    for (int i = 0; i < R; ++i) sc[i] = expf(logf(0.25f) + (logf(4.f) - logf(0.25f)) * (float)i / (float)(R - 1));
    float w0 = hW[0];
    void *dW, *dB, *dG, *dGb, *dH;
    HK(hipMalloc(&dW, sizeof(float) * NW)); HK(hipMalloc(&dB, sizeof(float) * NB)); HK(hipMalloc(&dG, sizeof(float) * NW));
    HK(hipMalloc(&dGb, sizeof(float) * NB)); HK(hipMalloc(&dH, sizeof(float) * NW));
    HK(hipMemcpy(dW, hW, sizeof(float) * NW, hipMemcpyHostToDevice)); HK(hipMemcpy(dB, hB, sizeof(float) * NB, hipMemcpyHostToDevice));
    const void* grads[2] = {dG, dGb};
    void* params[2] = {dW, dB};

    double m_h = 0.0, m_g = 0.0;                               /* averages over the last ten steps */
    for (int t = 0; t < STEPS; ++t) {
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < Cc; ++j) hG[i * Cc + j] = sc[i] * (float)nrand();
        for (int j = 0; j < NB; ++j) hGb[j] = (float)nrand();
        HK(hipMemcpy(dG, hG, sizeof(float) * NW, hipMemcpyHostToDevice)); HK(hipMemcpy(dGb, hGb, sizeof(float) * NB, hipMemcpyHostToDevice));
        const float beta = 0.f;                                 /* no momentum smoothing: whiten the raw gradient */
        CK(psgdk_accumulate(plan, grads, PSGDK_F32, NULL, PSGDK_F32, 0.f, beta, 1, NULL, NULL));
        CK(psgdk_update_precond_q0p5eq1p5(plan, PSGDK_SRC_GRAD, 0.3f, 0.9f, 1e-9f, NULL, 1234u, (uint64_t)t, NULL, NULL));
        CK(psgdk_precond_grad(plan, PSGDK_SRC_GRAD, NULL));
        CK(psgdk_read_precond_grad(plan, 0, dH, PSGDK_F32, 0, 2.f, 10.f, NULL));
        CK(psgdk_apply_update(plan, params, PSGDK_F32, 1e-3f, 0.f, 2.f, 10.f, NULL));
        HK(hipDeviceSynchronize());
        HK(hipMemcpy(hH, dH, sizeof(float) * NW, hipMemcpyDeviceToHost));
        for (int i = 0; i < NW; ++i) if (!isfinite(hH[i])) { fprintf(stderr, "non-finite preconditioned gradient at step %d\n", t); return 1; }
        if (t >= STEPS - 10) {
            /* || C - tr(C)/R I ||_F / tr(C) of the row covariance estimate C = x x^T, for x = h and x = g */
            for (int which = 0; which < 2; ++which) {
                const float* x = which ? hG : hH;
                double tr = 0, fro = 0;
                for (int i = 0; i < R; ++i) { double s = 0; for (int j = 0; j < Cc; ++j) s += (double)x[i * Cc + j] * x[i * Cc + j]; tr += s; }
                for (int i = 0; i < R; ++i)
                    for (int k = 0; k < R; ++k) {
                        double s = 0;
                        for (int j = 0; j < Cc; ++j) s += (double)x[i * Cc + j] * x[k * Cc + j];
                        if (i == k) s -= tr / R;
                        fro += s * s;
                    }
                *(which ? &m_g : &m_h) += sqrt(fro) / tr / 10.0;
            }
        }
    }
    HK(hipMemcpy(hW, dW, sizeof(float) * NW, hipMemcpyDeviceToHost));
    int64_t fb = -1;
    CK(psgdk_plan_info(plan, PSGDK_INFO_NLB_FALLBACKS, &fb));
    printf("abi_demo: distance of the row covariance from a multiple of I: gradient %.3f, preconditioned gradient %.3f (last 10 of %d "
           "steps); w[0] %.5f -> %.5f; norm-bound fallbacks %lld\n", m_g, m_h, STEPS, w0, hW[0], (long long)fb);
    if (!(m_h < 0.75 * m_g)) { fprintf(stderr, "the preconditioner did not whiten the gradient\n"); return 1; }
    if (hW[0] == w0 || !isfinite(hW[0])) { fprintf(stderr, "parameters did not move\n"); return 1; }
    CK(psgdk_plan_destroy(plan));
    (void)hipFree(state); (void)hipFree(work); (void)hipFree(dW); (void)hipFree(dB); (void)hipFree(dG); (void)hipFree(dGb); (void)hipFree(dH);
    puts("abi_demo ok");
    return 0;
}
