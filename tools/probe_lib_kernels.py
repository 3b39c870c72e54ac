"""Which hipBLASLt (Tensile) kernels the library picks for the compute-bound linear shapes where it is ahead of k_gemm_dma
(run under `rocprofv3 --kernel-trace --stats`; the kernel names carry the macro tile, the MFMA instruction and the LDS /
prefetch settings): tools/probe_lib_kernels.py"""
import torch
import torch.nn.functional as F

for (M, N, K) in ((32768, 3840, 1280), (32768, 10240, 1280), (131072, 5120, 640), (8192, 3840, 1280), (524288, 320, 320)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    for _ in range(3):
        y = F.linear(x, w, b)
    torch.cuda.synchronize()
    print(M, N, K, flush=True)
