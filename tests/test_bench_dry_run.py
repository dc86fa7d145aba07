"""CPU: bench.py's multi-rank host path at world size 2 over gloo (`--dry-run`: kernels stood in by host arithmetic).  Walks the
rendezvous, the native communicator's id shipping, the barriers, the all-gather of one partial per rank, the MAX / MIN reductions
and the rank-0 report - what the driver's 1/2/4/8-GPU scaling run exercises (reference collective:
/root/reference/src/core/distributed/distributed_context.cpp:97-122)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [1, 2])
def test_bench_dry_run_walks_every_collective(world):
    cmd = [sys.executable]
    if world > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--dry-run", "--native-comm"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stdout + run.stderr
    r = _line(run.stdout)
    assert r["dry_run"] is True and r["n_gpus"] == world and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["reduce_consistent"] is True and r["global_sum_matches_world1"] is True and r["native_comm_id_shipped"] is True
    assert r["collective_ab"]["allreduce_total_equals_allgather_total"] is True   # SURVEY.md 8(e)'s alternative exchange gives the same words
    assert r["per_rank_ct_mul_per_s"]["ranks"] == world and 0 < r["per_rank_ct_mul_per_s"]["min"] <= r["per_rank_ct_mul_per_s"]["max"]
    assert r["allgather_us"]["min"] <= r["allgather_us"]["median"] <= r["allgather_us"]["max"]
    assert r["config"]["global_batch"] == world * r["config"]["batch_per_gpu"]
    if world > 1:   # rank 0's C++ multi-process programs (stood in here) while the other ranks wait on the store, not in a collective
        progs = r["multi_gpu_programs"]
        assert progs["configs3_cpp_host"]["world"] == world and progs["configs3_cpp_host"]["matches_world1_recomputation"] is True
        assert progs["configs4_token_sharded_block"]["ranks"] == world and progs["configs4_token_sharded_block"]["tokens"] == 8 * world
        assert progs["configs4_single_token_tensor_parallel_ffn"]["world"] == world
    else:
        assert "multi_gpu_programs" not in r
    # value is the whole-job aggregate: all ranks' pairs over the MAX-over-ranks time
    assert abs(r["value"] - r["config"]["global_batch"] * r["steps"] / (r["ms_per_step"] * 1e-3 * r["steps"])) < 1e-6 * r["value"]


def test_bench_refuses_a_world_size_mismatch():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert run.returncode != 0 and "torch.distributed.run" in (run.stdout + run.stderr)


def test_run_program_kills_the_whole_process_group_on_timeout(tmp_path):
    """bench.run_program: the C++ multi-GPU programs fork one process per GPU; a timeout has to take the children down with the parent."""
    import time

    sys.path.insert(0, ROOT)
    import bench
    pidfile = tmp_path / "child.pid"
    script = tmp_path / "forker.sh"
    script.write_text(f"#!/bin/bash\nsleep 300 &\necho $! > {pidfile}\necho '{{\"started\": true}}'\nsleep 300\n")
    script.chmod(0o755)
    t0 = time.time()
    rc, text = bench.run_program([str(script)], 1.5)   # an absolute path replaces examples/<name>
    assert rc == -9 and "killed after" in text and '"started"' in text and time.time() - t0 < 20
    child = int(pidfile.read_text())
    for _ in range(50):
        try:
            os.kill(child, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the forked child survived the timeout")
    rc, text = bench.run_program(["/bin/echo", "{\"ok\": 1}"], 5)
    assert rc == 0 and text.strip() == '{"ok": 1}'
