import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
for M, N, K in [(50432, 768, 768), (50432, 2304, 768), (50432, 3072, 768), (5120, 6144, 2048)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.5).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    for _ in range(5):
        F.linear(a, w, b)
torch.cuda.synchronize()
