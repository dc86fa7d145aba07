"""Does the Infinity Cache keep freshly WRITTEN data?  (tool)  Writes a buffer, reads it back at once and after 4 GiB of other writes.
Measured on MI355X: up to 256 MiB the immediate read-back is 1.4-1.8x faster, beyond 256 MiB there is no difference - the cache is
memory-side and allocates on writes, so a consumer that follows a producer within 256 MiB reads from it."""
import torch, time
dev = torch.device("cuda", 0)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]
big = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
for mb in (32, 64, 128, 192, 256, 384, 512, 1024):
    n = mb << 20
    x = torch.empty(n // 8, dtype=torch.int64, device=dev)
    # read after write of the same buffer (fresh in whatever cache keeps writes)
    def fresh():
        x.fill_(3)
        return x.sum()
    def fill_only():
        x.fill_(3)
    def stale():
        x.fill_(3)
        big.fill_(1)      # 4 GiB of other traffic in between
        return x.sum()
    def big_only():
        big.fill_(1)
    tf, tw, ts_, tb = t(fresh), t(fill_only), t(stale), t(big_only)
    print(f"{mb:5d} MiB: write {tw:8.1f} us ({n / tw / 1e6:6.2f} TB/s)   read-after-write {tf - tw:8.1f} us ({n / max(tf - tw, 1e-3) / 1e6:6.2f} TB/s)   read after 4 GiB of other writes {ts_ - tw - tb:8.1f} us ({n / max(ts_ - tw - tb, 1e-3) / 1e6:6.2f} TB/s)")
