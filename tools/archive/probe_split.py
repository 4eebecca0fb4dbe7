"""GPU probe: v_dot2c_f32_bf16 residuals vs the shift/sub split (bit-exact?)."""
import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, 'build', 'probe_split.so'))
lib.probe_split.argtypes = [C.c_void_p] * 5 + [C.c_long, C.c_void_p]
dev = torch.device('cuda:0')
n = 1 << 22
g = torch.Generator().manual_seed(0)
parts = [torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-20, torch.randn(n, generator=g) * 1e20,
         torch.randn(n, generator=g) * 1e-38, torch.randint(-2**31, 2**31 - 1, (n,), generator=g, dtype=torch.int64).to(torch.int32).view(torch.float32)]
for name, x in zip(['N(0,1)', '1e-20', '1e20', '1e-38 (denormal residuals)', 'random bits'], parts):
    x = torch.nan_to_num(x, nan=1.0, posinf=2.0, neginf=-2.0).to(dev)
    ref = torch.empty(n, dtype=torch.int32, device=dev); alt = torch.empty_like(ref)
    r3 = torch.empty(n // 2, dtype=torch.int32, device=dev); a3 = torch.empty_like(r3)
    rc = lib.probe_split(x.data_ptr(), ref.data_ptr(), alt.data_ptr(), r3.data_ptr(), a3.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print('%-28s rc=%d  plane2 mismatches %d / %d, plane3 mismatches %d' % (name, rc, int((ref != alt).sum()), n, int((r3 != a3).sum())))
