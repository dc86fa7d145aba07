"""HBM traffic of the fused multiply measured NOW, on this box (tool; bench.py calls it after its timed region for `roofline.traffic`).

Driver mode (no arguments): runs itself twice under `rocprofv3 --pmc <one counter>` (FETCH_SIZE, then WRITE_SIZE: separate passes, no trace
domains, as MI355X_MICROARCH.md prescribes), reads the per-dispatch averages of the multiply's launches from the rocpd database and prints
one JSON line: bytes per ct-mul = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / pairs  (gfx950: FETCH_SIZE counts 128-byte requests at 64 B).
Worker mode (`worker <pairs>`): three launches of dpfhe_ct_mul over `pairs` ciphertext pairs at N=4096, L=4 (the probe of the forms is off)."""
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = 2048


def worker(pairs):
    sys.path.insert(0, ROOT)
    import torch
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
    from deeppowers_amd.params import FheParams
    p = FheParams.n4096_l4()
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    form = os.environ.get("DPFHE_LIVE_FORM")
    if form:
        ctx.set_ct_mul_variant(form)
    L, N, dev = p.n_limbs, p.n, ctx.device
    q = torch.tensor(p.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
    g = torch.Generator(device=dev).manual_seed(9)
    a = Ciphertext(torch.randint(0, 2**62, (pairs, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    b = Ciphertext(torch.randint(0, 2**62, (pairs, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    o = ctx.empty(pairs, components=3)
    for _ in range(3):
        ev.multiply(a, b, out=o)
    torch.cuda.synchronize()
    ctx.close()


def counter_avg(db_path, counter, grid):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    grid_col = next((c for c in ("grid_size", "grid_size_x", "grid_x") if c in cols), None)
    per = {}
    for name, cname, value, disp, gsz in cur.execute(f"select {name_col}, counter_name, value, dispatch_id, {grid_col} from counters_collection"):
        if cname == counter and "ct_mul_" in name and int(gsz) == grid:
            per[disp] = per.get(disp, 0.0) + value
    return (sum(per.values()) / len(per), len(per)) if per else (None, 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        return worker(int(sys.argv[2]))
    exe = shutil.which("rocprofv3")
    if not exe:
        print(json.dumps({"error": "rocprofv3 not on PATH"}))
        return
    form = sys.argv[1] if len(sys.argv) > 1 else ""
    env = dict(os.environ, TMPDIR="/tmp", DPFHE_LIVE_FORM=form)
    out = {"pairs": PAIRS, "form": form or "default"}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="dpfhe_pmc_", dir="/tmp")
        try:
            run = subprocess.run([exe, "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "worker", str(PAIRS)],
                                 capture_output=True, text=True, timeout=75, env=env, cwd="/tmp")
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if run.returncode != 0 or not dbs:
                out["error"] = f"{counter}: rc {run.returncode}: " + (run.stderr or run.stdout)[-200:]
                break
            avg, n = counter_avg(dbs[0], counter, PAIRS * 4 * 256)
            if avg is None:
                out["error"] = f"{counter}: no ct_mul dispatch of {PAIRS} pairs in the profile"
                break
            out[counter.lower() + "_kib"] = avg
            out["dispatches"] = n
        except Exception as e:
            out["error"] = f"{counter}: {e!r}"[:200]
            break
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "error" not in out:
        out["hbm_bytes_per_ct_mul"] = (2 * out["fetch_size_kib"] + out["write_size_kib"]) * 1024 / PAIRS
        out["algorithmic_bytes_per_ct_mul"] = 7 * 4 * 4096 * 8
    print(json.dumps(out))


if __name__ == "__main__":
    main()
