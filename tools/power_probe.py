"""Board power, power cap and shader clock while the NTT and the multiply kernels run back to back (tool, not product).

A sampler thread reads the amdgpu hwmon files (power1_average / power1_input, power1_cap, freq1_input) every 20 ms while the main
thread keeps ~2 s of launches queued per phase; `rocm-smi` is only used for the one-off static dump.  Output: one line per phase
with the mean / max of the samples taken while the queue was full.  This is the evidence for MEASUREMENTS.md section 5's statement that
the transforms run at a power-limited clock: the board sits at its cap and the clock settles below nominal.
"""
import glob, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams


def hwmon_dir():
    """The hwmon directory of HIP device 0 (a box may expose several cards in sysfs and only one to HIP): match the PCI address."""
    p = torch.cuda.get_device_properties(0)
    want = None
    if hasattr(p, "pci_bus_id"):
        want = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}."
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if not any(os.path.exists(os.path.join(d, f)) for f in ("power1_average", "power1_input")):
            continue
        addr = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
        cands.append((d, addr))
    print("# HIP device 0 PCI address:", want, " sysfs cards:", ", ".join(a for _, a in cands))
    for d, addr in cands:
        if want and addr.startswith(want):
            return d
    return cands[0][0] if len(cands) == 1 else None


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, d):
        super().__init__(daemon=True)
        self.d, self.rows, self.on, self.stop = d, [], False, False
        self.power_file = next((p for p in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, p))), None)

    def run(self):
        while not self.stop:
            if self.on:
                p = read_int(os.path.join(self.d, self.power_file)) if self.power_file else None
                f = read_int(os.path.join(self.d, "freq1_input"))
                self.rows.append((p, f))
            time.sleep(0.02)

    def take(self):
        rows, self.rows = self.rows, []
        return rows


def summarize(name, rows, cap_w, extra=""):
    ps = [p / 1e6 for p, _ in rows if p]
    fs = [f / 1e6 for _, f in rows if f]
    pw = f"power mean {sum(ps) / len(ps):7.1f} W  max {max(ps):7.1f} W" if ps else "power n/a"
    fr = f"sclk mean {sum(fs) / len(fs):6.0f} MHz  min {min(fs):6.0f} MHz" if fs else "sclk n/a"
    cap = f"cap {cap_w:6.0f} W" if cap_w else "cap n/a"
    print(f"{name:44s} {len(rows):4d} samples  {pw}  {cap}  {fr}  {extra}", flush=True)


def phase(sampler, name, fn, seconds, cap_w, units=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    sampler.take()
    sampler.on = True
    s.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()   # keeps the queue bounded; 20 launches = several ms of work, the gap is < 1 %
    e.record()
    torch.cuda.synchronize()
    sampler.on = False
    us = s.elapsed_time(e) * 1e3 / n
    summarize(name, sampler.take(), cap_w, f"{us:8.1f} us/launch" + (f" = {units(us)}" if units else ""))


def main():
    d = hwmon_dir()
    print("# hwmon:", d)
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks"], capture_output=True, text=True, timeout=60).stdout
        print("\n".join("# " + l for l in out.splitlines() if l.strip() and "====" not in l))
    except Exception as ex:
        print("# rocm-smi unavailable:", ex)
    if d is None:
        print("no hwmon directory with power files: nothing to sample")
        return
    cap = read_int(os.path.join(d, "power1_cap"))
    cap_w = cap / 1e6 if cap else None
    sampler = Sampler(d)
    sampler.start()

    sampler.on = True
    time.sleep(1.0)
    sampler.on = False
    summarize("idle", sampler.take(), cap_w)

    params = FheParams.n4096_l4()
    ctx = Context(params, 0)
    ev = Evaluator(ctx)
    L, N = params.n_limbs, params.n
    q = torch.tensor(params.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
    g = torch.Generator(device=ctx.device).manual_seed(5)
    nb = 8192
    x = torch.randint(0, 2**62, (nb, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    nbytes = 2 * N * 8 * nb * L
    hbm = lambda us: f"{nbytes / us / 8e6 * 100:5.1f} % of 8 TB/s"
    phase(sampler, "forward NTT, 8192 x 4 polys per launch", lambda: ev.ntt_forward(x, out=y), 3.0, cap_w, hbm)
    phase(sampler, "inverse NTT, 8192 x 4 polys per launch", lambda: ev.ntt_inverse(x, out=y), 3.0, cap_w, hbm)
    phase(sampler, "device copy of the same bytes (torch)", lambda: y.copy_(x), 3.0, cap_w, hbm)
    cb = 8192
    a = Ciphertext(torch.cat([x.view(-1, 2, L, N)] * 2))
    b = Ciphertext(torch.cat([y.view(-1, 2, L, N)] * 2))
    o = ctx.empty(cb, components=3)
    phase(sampler, "ct_mul, 8192 pairs per launch", lambda: ev.multiply(a, b, out=o), 3.0, cap_w, lambda us: f"{cb / us:6.3f} M ct-mul/s")
    sampler.stop = True


if __name__ == "__main__":
    main()
