"""Per-item timeline of the 256x256 GEMM kernel from the instrumented build (make -C youku-mplug_amd/csrc timing;
MPV_LIB_PATH=youku-mplug_amd/csrc/build/libmpv_hip_timing.so python tools/probe/gemm256_timeline.py M N K [ta tb]).
Stamps (shader clocks, wave 0 of each workgroup): kernel start, after the prologue barrier, then per item:
item start, main loop done, ring retired + un-stagger, epilogue done."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import youku_mplug_amd
from youku_mplug_amd import ops, _lib

dev = torch.device("cuda:0")


def run(M, N, K, ta=0, tb=0):
    a = ((torch.rand(K, M, device=dev) if ta else torch.rand(M, K, device=dev)) * 2 - 1).bfloat16()
    b = ((torch.rand(K, N, device=dev) if tb else torch.rand(N, K, device=dev)) * 2 - 1).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        ops.gemm(a, b, M, N, K, out=out, trans_a=bool(ta), trans_b=bool(tb), tile_hint=256)
    torch.cuda.synchronize()
    buf = (C.c_longlong * (256 * 64))()
    fn = _lib.lib().mpv_gemm256_read_timeline
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
    assert fn(buf) == 0          # clears the device buffer
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.gemm(a, b, M, N, K, out=out, trans_a=bool(ta), trans_b=bool(tb), tile_hint=256)
    e.record()
    torch.cuda.synchronize()
    assert fn(buf) == 0
    t = np.frombuffer(buf, dtype=np.int64).reshape(256, 64)
    act = t[:, 60] != 0
    t = t[act]
    wall = (t[:, 60] - t[:, 62]) / 100.0     # us (100 MHz)
    clk = (t[:, 61] - t[:, 63])
    mhz = np.median(clk / np.maximum(wall, 1e-9))
    print(f"M={M} N={N} K={K} ta={ta} tb={tb}: event time {s.elapsed_time(e)*1e3:.1f} us; workgroup lifetime median {np.median(wall):.1f} us "
          f"(min {wall.min():.1f} max {wall.max():.1f}); shader clock ~{mhz:.0f} MHz")
    items = 0
    # slots: 0 start, 1 after prologue, then 4 per item
    d = t[:, :60].astype(np.float64)
    pro = (d[:, 1] - d[:, 0]) / mhz
    print(f"  prologue (setup + first K-tile landed): median {np.median(pro):.2f} us  max {pro.max():.2f}")
    for it in range(14):
        b0 = 2 + 4 * it
        if b0 + 3 >= 60:
            break
        valid = (d[:, b0 + 3] > 0) & (d[:, b0] > 0)
        if valid.sum() == 0:
            break
        x = d[valid]
        main = (x[:, b0 + 1] - x[:, b0]) / mhz
        ret = (x[:, b0 + 2] - x[:, b0 + 1]) / mhz
        epi = (x[:, b0 + 3] - x[:, b0 + 2]) / mhz
        print(f"  item {it}: {valid.sum():3d} wgs  main loop {np.median(main):6.2f} us (max {main.max():6.2f})  retire+unstagger {np.median(ret):5.2f}  "
              f"epilogue {np.median(epi):5.2f} (max {epi.max():5.2f})")


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        run(*[int(v) for v in sys.argv[1:]])
    else:
        for shp in [(50432, 2304, 768), (50432, 768, 768), (5120, 2048, 2048), (5120, 6144, 2048), (8192, 8192, 8192), (2304, 768, 50432, 1, 1)]:
            run(*shp)
