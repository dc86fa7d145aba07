"""Which clock domain differs between the two regimes of the multiply (MEASUREMENTS.md section 5)?  (tool)
Phases of ~2 s each: the bench step from a rested chip, a 1 GiB copy loop, the bench step again; a sampler thread reads the `*`-marked level
of every pp_dpm_* file of the device and the hwmon power / sclk every 10 ms.  Prints, per phase, the step time and the mean of each clock."""
import glob, os, re, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams
from deeppowers_amd.sharding import ShardedMultiplyReduce
from tools.power_probe import hwmon_dir, read_int

hw = hwmon_dir()
dev_dir = os.path.realpath(os.path.join(hw, "..", "..")) if hw else None
dpm_files = sorted(glob.glob(os.path.join(dev_dir, "pp_dpm_*"))) if dev_dir else []
print("# device dir:", dev_dir, " dpm files:", [os.path.basename(f) for f in dpm_files])


def current_level(path):
    try:
        for line in open(path):
            if "*" in line:
                m = re.search(r"(\d+)\s*Mhz", line, re.I)
                return int(m.group(1)) if m else None
    except Exception:
        return None
    return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.on, self.stop = [], False, False

    def run(self):
        while not self.stop:
            if self.on:
                row = {os.path.basename(f): current_level(f) for f in dpm_files}
                row["power_w"] = (read_int(os.path.join(hw, "power1_average")) or read_int(os.path.join(hw, "power1_input")) or 0) / 1e6
                row["hwmon_sclk"] = (read_int(os.path.join(hw, "freq1_input")) or 0) / 1e6
                row["hwmon_mclk"] = (read_int(os.path.join(hw, "freq2_input")) or 0) / 1e6
                self.rows.append(row)
            time.sleep(0.01)


p = FheParams.n4096_l4()
ctx = Context(p, 0); ev = Evaluator(ctx)
L, N, B = p.n_limbs, p.n, 8192
q = torch.tensor(p.moduli, dtype=torch.int64, device=ctx.device).view(1, 1, L, 1)
a = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), dtype=torch.int64, device=ctx.device) % q)
b = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), dtype=torch.int64, device=ctx.device) % q)
pipe = ShardedMultiplyReduce(ev, B)
src = torch.randint(0, 2**62, (1 << 27,), dtype=torch.int64, device=ctx.device)
dst = torch.empty_like(src)
x = torch.randint(0, 2**62, (1024, L, N), dtype=torch.int64, device=ctx.device) % q.view(1, L, 1)
smp = Sampler(); smp.start()


def phase(name, fn, seconds, unit):
    fn(); torch.cuda.synchronize()
    smp.rows = []; smp.on = True
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        torch.cuda.synchronize(); n += 8
    dt = time.perf_counter() - t0
    smp.on = False
    rows = smp.rows
    keys = sorted(rows[0].keys()) if rows else []
    means = {k: sum(r[k] for r in rows if r[k] is not None) / max(1, sum(1 for r in rows if r[k] is not None)) for k in keys}
    print(f"{name:34s} {dt / n * 1e3:8.3f} ms/{unit}  " + "  ".join(f"{k.replace('pp_dpm_', '')}={v:.0f}" for k, v in means.items()))


time.sleep(3)   # rested chip
phase("multiply step, rested chip", lambda: pipe.step(a, b), 2.0, "step")
phase("multiply step, continued", lambda: pipe.step(a, b), 2.0, "step")
phase("copy 1 GiB (dpfhe_copy)", lambda: ev.device_copy(src, dst), 2.0, "copy")
phase("multiply step, after the copies", lambda: pipe.step(a, b), 0.3, "step")
phase("multiply step, continued", lambda: pipe.step(a, b), 2.0, "step")
phase("forward NTT configs[1] in place", lambda: ev.ntt_forward_(x), 2.0, "launch")
phase("multiply step, after the NTTs", lambda: pipe.step(a, b), 0.3, "step")
phase("multiply step, continued", lambda: pipe.step(a, b), 2.0, "step")
smp.stop = True
