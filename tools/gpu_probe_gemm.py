"""Quick GPU probe of the GEMM kernel + Philox stream (pre-pytest bring-up)."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "psgd_torch_amd", "libpsgdk.so"))
L.psgdk_test_gemm_nt.argtypes = [C.c_void_p]*4 + [C.c_int]*9 + [C.c_void_p]
L.psgdk_fill_normal.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
ok = True
def stream(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
for dt, code, tol in ((torch.bfloat16, 0, 2e-2), (torch.float32, 1, 1e-5)):
    for (M, N, K) in ((128,128,64), (64,64,64), (192,320,128), (768,768,768), (2304,768,768), (64,192,1024)):
        A = torch.randn(M, K, device=dev).to(dt); B = torch.randn(N, K, device=dev).to(dt)
        Cc = torch.zeros(M, N, device=dev, dtype=dt); Ct = torch.zeros(N, M, device=dev, dtype=dt)
        st = L.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), Ct.data_ptr(), code, M, N, K, K, K, N, M, 0, stream())
        torch.cuda.synchronize()
        ref = (A.double() @ B.double().t())
        e1 = ((Cc.double()-ref).norm()/ref.norm()).item(); e2 = ((Ct.double().t()-ref).norm()/ref.norm()).item()
        good = st == 0 and e1 < tol and e2 < tol
        ok &= good
        print(f"gemm {dt} {M}x{N}x{K} status={st} relerr C={e1:.2e} Ct={e2:.2e} {'OK' if good else 'FAIL'}")
    # symmetric
    for (M, K) in ((128, 64), (192, 256), (768, 2304)):
        A = torch.randn(M, K, device=dev).to(dt)
        Cc = torch.full((M, M), float('nan'), device=dev, dtype=dt)
        st = L.psgdk_test_gemm_nt(A.data_ptr(), A.data_ptr(), Cc.data_ptr(), None, code, M, M, K, K, K, M, M, 1, stream())
        torch.cuda.synchronize()
        ref = A.double() @ A.double().t()
        e1 = ((Cc.double()-ref).norm()/ref.norm()).item()
        symm = torch.equal(Cc, Cc.t())
        good = st == 0 and e1 < tol and symm
        ok &= good
        print(f"syrk {dt} {M}x{K} status={st} relerr={e1:.2e} bitwise_sym={symm} {'OK' if good else 'FAIL'}")
# padded leading dims
A = torch.randn(128, 256, device=dev).to(torch.bfloat16); B = torch.randn(192, 256, device=dev).to(torch.bfloat16)
Cc = torch.zeros(128, 256, device=dev, dtype=torch.bfloat16)
st = L.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, 0, 128, 192, 128, 256, 256, 256, 0, 0, stream())
torch.cuda.synchronize()
ref = A[:, :128].double() @ B[:, :128].double().t()
e = ((Cc[:, :192].double()-ref).norm()/ref.norm()).item(); z = float(Cc[:, 192:].abs().max())
print(f"gemm ld-padded relerr={e:.2e} untouched_pad={z} {'OK' if e<2e-2 and z==0 else 'FAIL'}"); ok &= (e < 2e-2 and z == 0)
# philox
x = torch.empty(1<<22, device=dev, dtype=torch.float32)
L.psgdk_fill_normal(x.data_ptr(), 1, x.numel(), 1234, 0, 7, stream()); torch.cuda.synchronize()
m, v, k4 = x.mean().item(), x.var().item(), (x**4).mean().item()
print(f"philox normal mean={m:.4f} var={v:.4f} kurt={k4:.3f} min={x.min().item():.2f} max={x.max().item():.2f}")
ok &= abs(m) < 5e-3 and abs(v-1) < 1e-2 and abs(k4-3) < 0.05
y = torch.empty(1<<22, device=dev, dtype=torch.float32)
L.psgdk_fill_normal(y.data_ptr(), 1, y.numel(), 1234, 0, 8, stream()); torch.cuda.synchronize()
print("stream corr", (x*y).mean().item()); ok &= abs((x*y).mean().item()) < 5e-3
# timing
for dt, code in ((torch.bfloat16, 0), (torch.float32, 1)):
    M, N, K = 8192, 768, 768
    A = torch.randn(M, K, device=dev).to(dt); B = torch.randn(N, K, device=dev).to(dt); Cc = torch.zeros(M, N, device=dev, dtype=dt)
    for _ in range(3): L.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, code, M, N, K, K, K, N, M, 0, stream())
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): L.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, code, M, N, K, K, K, N, M, 0, stream())
    torch.cuda.synchronize(); dtm = (time.time()-t0)/20
    print(f"timing (incl malloc/sync overhead) {dt} {M}x{N}x{K}: {dtm*1e6:.0f} us -> {2*M*N*K/dtm/1e12:.1f} TF")
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
