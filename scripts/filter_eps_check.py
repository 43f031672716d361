"""How far the f16 matrix-core dot product of flat_filter.hip is from the exact one, against the bound the gate uses.

The library exposes the filter only through its answers, so this measures the arithmetic directly with torch on the same
hardware path (f16 inputs rounded to nearest even, MFMA f16 -> f32 accumulate via torch.matmul on half tensors with f32
output is not available; the product is formed as (x16.float() @ q16.float().T) in f32 as the reference of what an f32
accumulation of exact f16 products gives, and as torch.matmul(x16, q16.T) -- the hardware f16 MFMA path -- for the value
the kernel's matrix cores produce up to accumulation order), and compares both with the f64 dot product of the f32 inputs."""
import sys
import numpy as np
import torch

dev = torch.device("cuda", 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
g = torch.Generator(device=dev)
g.manual_seed(1)
worst = 0.0
for trial in range(4):
    A = torch.randn(D, 32, generator=g, device=dev)
    x = torch.nn.functional.normalize(torch.randn(200_000, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(200_000, D, generator=g, device=dev), dim=1)
    q = torch.nn.functional.normalize(torch.randn(256, 32, generator=g, device=dev) @ A.T + 0.05 * torch.randn(256, D, generator=g, device=dev), dim=1)
    exact = (x.double() @ q.double().T)
    x16, q16 = x.half(), q.half()
    approx_f32acc = x16.float() @ q16.float().T
    approx_mfma = torch.matmul(x16, q16.T).float()          # f16 output: adds an output rounding the kernel does not have
    e1 = (approx_f32acc.double() - exact).abs().max().item()
    e2 = (approx_mfma.double() - exact).abs().max().item()
    worst = max(worst, e1)
    print(f"trial {trial}: max |f16 products, f32 accumulate - exact| = {e1:.3e}   (f16-output matmul: {e2:.3e})")
rel = 2 ** -10 + 2 ** -22 + D * 2 ** -22 + (D / 16 + 5) * 2 ** -24
eps = (1.0002 * rel + 1.004 * 2 ** -25 * D ** 0.5 * 2 + 2 ** -23 * 2) * 1.001
print(f"gate margin eps for unit vectors at D={D}: {eps:.3e}; worst observed error {worst:.3e} = {worst / eps:.3f} of it")
