"""Calibration: what fp32 GEMM rate does the vendor library (rocBLAS/hipBLASLt via torch.matmul) reach on this
device for the conv kernel's GEMM shapes?  (A known-good reference on the same hardware.)"""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
for (M, K, N) in [(107008, 4608, 256), (107008, 2304, 128), (26752, 8064, 512), (428032, 1152, 128), (1712128, 1152, 64),
                  (8192, 8192, 8192), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(K, N, device="cuda")
    c = a @ b
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record(); c = a @ b; e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    print("M=%8d K=%6d N=%5d  %9.1f us  %6.1f TF" % (M, K, N, best * 1e3, 2.0 * M * K * N / best / 1e9), flush=True)
