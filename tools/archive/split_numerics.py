"""CPU emulation of split-precision products: error vs fp64 of (a) a plain fp32 GEMM, (b) bf16x6 (three bf16 planes, six
terms), (c) fp16x3 (two fp16 planes after a power-of-two scale, three terms), (d) bf16 two planes / three terms.
Products of the planes are exact in fp32 and accumulated in fp64 here, so only the REPRESENTATION + dropped-term error
shows (the MFMA's fp32 accumulation adds the same summation noise to all of them).  python tools/split_numerics.py"""
import torch
torch.manual_seed(0)
M, K, N = 512, 1152, 128


def planes_bf16(x, n):
    out, r = [], x.clone()
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p); r = r - p
    return out


def planes_f16(x, amax_bound):
    s = 2.0 ** torch.floor(torch.log2(torch.tensor(2.0 ** 14 / amax_bound)))
    xs = x * s
    h1 = xs.to(torch.float16).to(torch.float32)
    h2 = (xs - h1).to(torch.float16).to(torch.float32)
    return [h1, h2], float(s)


def report(name, got, ref):
    err = (got - ref).abs()
    scale = ref.abs().max()
    print('%-34s max err / max|ref| %.2e   rms err / rms ref %.2e' % (name, (err.max() / scale).item(), (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()))


for label, a, w in [
        ('relu(bn) activations x N(0,0.03) weights', torch.relu(torch.randn(M, K) * 1.3 + 0.2), torch.randn(N, K) * 0.03),
        ('wide-range operand (log-uniform 1e-6..1)', torch.exp(torch.rand(M, K) * 13.8 - 13.8) * torch.sign(torch.randn(M, K)), torch.randn(N, K) * 0.03),
        ('gradient-like 1e-5 * N(0,1) x activations', torch.randn(M, K) * 1e-5, torch.relu(torch.randn(N, K)))]:
    print(label)
    ref = a.double() @ w.double().t()
    report('fp32 operands, exact products', (a.double() @ w.double().t()).float().double(), ref)
    report('plain fp32 GEMM (torch CPU)', (a @ w.t()).double(), ref)
    A, W = planes_bf16(a, 3), planes_bf16(w, 3)
    six = sum(A[i].double() @ W[j].double().t() for i in range(3) for j in range(3) if i + j <= 2)
    report('bf16x6 (3 planes, 6 terms)', six, ref)
    three = sum(A[i].double() @ W[j].double().t() for i in range(2) for j in range(2) if i + j <= 1)
    report('bf16 2 planes, 3 terms', three, ref)
    for bound_mult, tag in ((1.0, 'exact amax'), (64.0, 'amax bound 64x loose')):
        (a1, a2), sa = planes_f16(a, a.abs().max().item() * bound_mult)
        (w1, w2), sw = planes_f16(w, w.abs().max().item() * bound_mult)
        f3 = (a1.double() @ w1.double().t() + a1.double() @ w2.double().t() + a2.double() @ w1.double().t()) / (sa * sw)
        report('fp16x3 (2 planes, 3 terms), ' + tag, f3, ref)
    print()
