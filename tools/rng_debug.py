import sys, torch, ctypes as C
sys.path.insert(0, '/root/repo')
from safe_denoiser_amd import _lib
from safe_denoiser_amd.rng import BatchedNormal
dev = torch.device('cuda', 0)
for numel in (16384, 1024, 5):
    g = torch.Generator(device=dev).manual_seed(42)
    off0 = g.get_offset()
    ref = torch.randn(numel, generator=g, device=dev)
    off1 = g.get_offset()
    gi, inc = C.c_int32(), C.c_int64()
    _lib.lib().sdn_randn_philox_plan(numel, C.byref(gi), C.byref(inc))
    out = torch.empty(1, numel, device=dev)
    meta = torch.tensor([[42], [off0]], dtype=torch.int64).to(dev)
    _lib.check(_lib.lib().sdn_randn_philox(meta[0].data_ptr(), meta[1].data_ptr(), None, 1, numel, out.data_ptr(), _lib.stream_ptr()), "x")
    torch.cuda.synchronize()
    d = (out[0] - ref).abs()
    print(numel, "torch offset", off0, "->", off1, "plan grid", gi.value, "inc", inc.value, "n_diff", int((out[0] != ref).sum()), "max abs diff", float(d.max()),
          "ref[:4]", ref[:4].tolist(), "mine[:4]", out[0, :4].tolist())
    # is ref a permutation / shifted version?
    if numel >= 1024:
        s = set(ref.tolist()); m = set(out[0].tolist())
        print("  common values:", len(s & m))
