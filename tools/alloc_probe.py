import torch, time
torch.cuda.init(); x = torch.empty(1, device="cuda"); torch.cuda.synchronize()
keep = []
for gb in (0.25, 0.9, 0.9, 0.9, 4.0, 0.9):
    t0 = time.perf_counter(); t = torch.empty(int(gb * 2**30 // 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("alloc %.2f GB: %.1f ms" % (gb, dt * 1e3)); keep.append(t)
del keep; torch.cuda.synchronize()
t0 = time.perf_counter(); t = torch.empty(int(0.9 * 2**30 // 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize(); print("re-alloc from cache 0.9 GB: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); torch.cuda.empty_cache(); torch.cuda.synchronize(); print("empty_cache: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
