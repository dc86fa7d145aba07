"""-m gpu: builds (g++) and runs the C++ facade test - host code in C++ calling HIP through the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_cpp_test():
    exe = os.path.join(ROOT, "tests", "cpp", "test_fhe_api")
    lib = os.path.join(ROOT, "deeppowers_amd")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    subprocess.check_call([
        "g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_fhe_api.cpp"),
        "-o", exe, "-L" + lib, "-ldpfhe_api", "-ldpfhe_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
        "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_facade_compiles_and_links():
    """CPU: the facade header, its library and the test program link (no GPU call)."""
    assert os.path.exists(build_cpp_test())


def test_cpp_prime_chain_is_the_python_generators(tmp_path):
    """CPU: FheParams::n8192(20) (the tabulated N = 8192 chain the multi-block example and its multiply workspaces are built on) equals
    params.ntt_primes(13, 20) - largest primes below 2^60 that are 1 mod 2N, smallest primitive 2N-th roots - and its first six are the pinned ones."""
    from deeppowers_amd.params import FheParams, ntt_primes
    src = tmp_path / "chain.cpp"
    src.write_text('#include <cstdio>\n#include <deeppowers/fhe.hpp>\nint main() { auto p = deeppowers::fhe::FheParams::n8192(20);\n'
                   'for (size_t i = 0; i < p.n_limbs(); ++i) std::printf("%llu %llu\\n", (unsigned long long)p.moduli[i], (unsigned long long)p.psi[i]); }\n')
    lib = os.path.join(ROOT, "deeppowers_amd")
    exe = str(tmp_path / "chain")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe, "-L" + lib, "-ldpfhe_api", "-ldpfhe_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])
    got = [tuple(int(v) for v in l.split()) for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()]
    want = ntt_primes(13, 20)
    assert got == list(zip(want.moduli, want.psi))
    assert tuple(q for q, _ in got[:6]) == FheParams.n8192_l6().moduli and FheParams.n8192(6) == FheParams.n8192_l6()


def build_example(name="encrypted_multiply"):
    """examples/Makefile (what __graft_entry__.build() runs): rebuilt only when the source or the library changed"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples"), name])
    return os.path.join(ROOT, "examples", name)


def test_example_compiles():
    assert all(os.path.exists(build_example(e)) for e in ("encrypted_multiply", "bench_ct_mul", "encrypted_linear", "encrypted_gpt2_linear", "encrypted_gpt2_ffn", "encrypted_gpt2_ffn_act", "encrypted_gpt2_block_act", "encrypted_gpt2_stack",
                                                          "encrypted_gpt2_block", "sharded_ct_mul", "sharded_ffn"))


@pytest.mark.gpu
def test_example_encrypted_linear_layer():
    out = subprocess.run([build_example("encrypted_linear")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("layer", ["qkv", "ffn_up", "ffn_down", "square"])
def test_example_encrypted_gpt2_layers_full_size(layer):
    """N3 at the reference's real shapes (gpt_model.cpp:793 QKV 768 -> 2304, :848 FFN 768 -> 3072 -> 768; attention output
    768 -> 768): N=8192, 5 data limbs + special prime; the decrypted, unpacked result must equal W x mod t in every output."""
    out = subprocess.run([build_example("encrypted_gpt2_linear"), layer, "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout and "MISMATCH" not in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_example_encrypted_gpt2_layers_eight_tokens_per_launch():
    """The same layers with 8 encrypted hidden states per application (keys and diagonals read once for all of them): every token's
    decrypted result equals W x mod t."""
    out = subprocess.run([build_example("encrypted_gpt2_linear"), "all", "1", "text", "8"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout and "MISMATCH" not in out.stdout and "8 token(s)" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("layer,tokens", [("qkv", 1), ("qkv", 8), ("ffn_down", 2)])
def test_example_encrypted_gpt2_layers_at_n16384(layer, tokens):
    """Round 5: the same layers at N = 16384 (six primes that are 1 mod 2^15: a 360-bit modulus under key switching, inside the 438-bit budget of 128-bit
    security at this ring degree).  No fused key-switch kernel exists there: the baby steps run on the stream kernels and the batched transforms, the giant
    steps' per-key inner products (dpfhe_switch_key_qp) and the fold rotations (dpfhe_rotate_hybrid_grouped) are composed from lift + batched transform +
    key_inner_product_kernel with one key per item group, scratch from the context's per-stream arena - in a process WITHOUT PyTorch (the regression case
    for the arena: a stream-ordered pool gave wrong words here).  Every token's decrypted result must equal W x mod t."""
    out = subprocess.run([build_example("encrypted_gpt2_linear"), layer, "1", "text", str(tokens), "14"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout and "MISMATCH" not in out.stdout and f"{tokens} token(s)" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_example_encrypted_gpt2_lm_head_tile():
    """gpt_model.cpp:883 logits: 768 -> 50257 is 7 output ciphertexts sharing the baby steps; a 3-ciphertext tile (768 -> 20000)
    runs here, the full head in examples/encrypted_gpt2_linear lm_head (2.3 GB of diagonals, ~1 min of host-side encoding)."""
    out = subprocess.run([build_example("encrypted_gpt2_linear"), "20000x768", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout and "3 output ciphertext" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_sharded_ct_mul_native_comm_world_size_one():
    """(e) in pure C++: one process per GPU, dpfhe_comm_* all-gather, id shipped through a file.  World size 1 here (one GPU
    on the test box); `sharded_ct_mul 8` is the 8-GPU form."""
    out = subprocess.run([build_example("sharded_ct_mul"), "1", "96", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_bench_runs():
    import json
    out = subprocess.run([build_example("bench_ct_mul"), "256", "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["host"] == "c++" and d["value"] > 1e4


@pytest.mark.gpu
def test_example_encrypt_multiply_relinearize_decrypt():
    out = subprocess.run([build_example()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_facade_parity_on_gpu():
    exe = build_cpp_test()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


@pytest.mark.gpu
def test_example_encrypted_ffn_block_chained_on_the_device():
    """Two packed layers chained under encryption at BASELINE configs[4]'s sizes (N=8192, 5 data limbs + special prime): W_up 768 -> 3072,
    hand-over to the next layer's input packing on the device (row swap + add), W_down 3072 -> 768, residual add; 4 tokens per
    application.  The program checks the hand-over slot by slot and the result against x + W_down (W_up x) mod t."""
    out = subprocess.run([build_example("encrypted_gpt2_ffn"), "4", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout and "MISMATCH" not in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_example_activated_ffn_block_uses_the_metric_op_and_switches_modulus():
    """y = x + W_down (W_up x)^2 over Z_65537 at the reference's FFN shapes (gpt_model.cpp:842-859 with the square standing in for GELU): W_up on five
    limbs, modulus switch to two, the activation as an EXACT ciphertext x ciphertext multiply (ExactMultiplier around the fused ct x ct kernel)
    + relinearisation, W_down on two limbs; every stage decrypted; the noise budget falls monotonically and stays positive."""
    import json
    out = subprocess.run([build_example("encrypted_gpt2_ffn_act"), "2", "1", "json"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-500:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["correct"] is True and d["activation_correct"] is True and d["ct_ct_multiplies_per_token"] == 1
    b = d["budget_bits"]
    assert b[0] > b[1] > 0 and b[2] > b[3] > b[4] > 0, b       # fresh > after W_up;  switched > squared > after W_down


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["flat", "ladder"])
def test_example_whole_block_with_activation(mode):
    """configs[4] as a forward pass of ONE block with its non-linearity: qkv -> v hand-over -> W_o + residual -> W_up -> modulus switch to 2 limbs ->
    square (exact ct x ct multiply) + relinearise -> W_down + residual; h1, the activation and h2 decrypted and compared; eight budget readings.
    `ladder`: the modulus falls with the budget inside the block (5 / 4 / 3 / 2 limbs) - same final budget, a sixth less time."""
    import json
    out = subprocess.run([build_example("encrypted_gpt2_block_act"), "2", "1", "json"] + (["ladder"] if mode == "ladder" else []), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-500:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    b = d["budget_bits"]
    assert d["correct"] is True and len(b) == 8 and b[0] > b[1] > b[2] > b[3] > b[4] > 0 and b[5] > b[6] > b[7] > 20, b


@pytest.mark.gpu
def test_example_whole_block_with_activation_on_a_128_bit_secure_ring():
    """Round 5: the same block at N = 16384 on six primes = 1 mod 2^15 - 360 bits under key switching, inside the 438 bits the Homomorphic Encryption
    Standard allows at 128-bit security for this ring degree (ternary secret, sigma 3.2) - with every key switch and the multiply composed from the
    batched transforms (no fused kernel above N = 8192): h1, the activation and h2 decrypt to the plaintext forward."""
    import json
    out = subprocess.run([build_example("encrypted_gpt2_block_act"), "2", "1", "json", "ladder", "14"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-500:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    b = d["budget_bits"]
    assert d["correct"] is True and d["log2_n"] == 14 and d["modulus_bits_under_key_switching"] <= d["he_standard_128bit_budget_bits"] == 438
    assert len(b) == 8 and b[0] > b[1] > b[2] > b[3] > b[4] > 0 and b[5] > b[6] > b[7] > 20, b


@pytest.mark.gpu
@pytest.mark.parametrize("log2n", [13, 14])
def test_example_whole_block_two_tokens_per_ciphertext(log2n):
    """Round 6: the two slot ROWS of a ciphertext carry two different tokens (PackedLinear / PackedSelect tokens_per_ciphertext = 2: every rotation of the
    diagonal method is a row rotation and the diagonals are the same for both rows, so the same kernels compute W x_A in row 0 and W x_B in row 1): the
    whole activated block on 4 tokens in 2 ciphertexts - h1, the activation and h2 of BOTH rows of every ciphertext decrypt to the plaintext forward, at
    N = 8192 and on the 128-bit-secure ring N = 16384; and a single packed layer (QKV: three output blocks in the windows of one row) the same way."""
    import json
    out = subprocess.run([build_example("encrypted_gpt2_block_act"), "4", "1", "json", "ladder", str(log2n), "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-500:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    b = d["budget_bits"]
    assert d["correct"] is True and d["log2_n"] == log2n and d["tokens"] == 4 and d["tokens_per_ciphertext"] == 2
    assert len(b) == 8 and b[0] > b[1] > b[2] > b[3] > b[4] > 0 and b[5] > b[6] > b[7] > 20, b
    for layer in ("qkv", "ffn_down"):
        out = subprocess.run([build_example("encrypted_gpt2_linear"), layer, "1", "json", "4", str(log2n), "2"], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-500:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert d["correct"] is True and d["tokens_per_ciphertext"] == 2 and d["tokens_per_apply"] == 4 and d["output_ciphertexts"] == 1
    if log2n == 13:   # ... and two blocks in a row on the modulus chain (the deeper demonstration), both rows of both blocks' outputs
        out = subprocess.run([build_example("encrypted_gpt2_stack"), "4", "1", "json", "7", "0", "2"], capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-500:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert d["correct"] is True and d["tokens_per_ciphertext"] == 2 and d["blocks"] >= 2 and d["correct_blocks"] == d["blocks"]


@pytest.mark.gpu
def test_example_two_blocks_on_a_modulus_chain():
    """configs[4] as a forward pass, DEEPER than one block: two blocks with the square activation chained on seven data limbs (420 bits); the limb
    count of every level is planned from a budget model (no secret key involved) and falls 7 -> 2 over the twelve levels; the first block's
    activation is an exact multiply at a five- or six-limb level (11 / 13-limb workspace); BOTH blocks' outputs decrypt to the plaintext forward."""
    import json
    out = subprocess.run([build_example("encrypted_gpt2_stack"), "2", "1", "json"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-1500:] + out.stderr[-500:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["correct"] is True and d["blocks"] == 2 and d["correct_blocks"] == 2 and d["ct_ct_multiplies_per_token"] == 2, d
    b = d["budget_bits"]
    lv = [int(v) for v in d["limbs_per_level"].replace("|", " ").split()]
    assert len(b) == 12 and all(x > y for x, y in zip(b, b[1:])) and b[-1] > 10, b
    assert len(lv) == 12 and lv[0] == 7 and lv[-1] == 2 and all(x >= y for x, y in zip(lv, lv[1:])), lv


@pytest.mark.gpu
def test_example_tensor_parallel_encrypted_ffn():
    """configs[4]'s shape: the encrypted FFN linear path sharded over the inner dimension, one process per GPU, partial ciphertexts
    all-gathered over the library's communicator and summed.  World size 1 runs the real multi-process program (fork, id through a
    file, RCCL); world sizes 2, 4 and 8 are played by one process on this box's single GPU (`emulate`: real slices, real hand-over
    between the two packed layers - a different one for each slice width -, real sum; the gather is a host copy).  Every run decrypts to
    W_down (W_up x) mod t and all of them print the same checksum."""
    import json
    sums = set()
    for args in (["1", "1"], ["2", "1", "0", "emulate"], ["4", "1", "0", "emulate"], ["8", "1", "0", "emulate"]):
        out = subprocess.run([build_example("sharded_ffn")] + args, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
        line = json.loads(next(l for l in out.stdout.splitlines() if l.startswith("{")))
        assert line["correct"] and line["world"] == int(args[0])
        sums.add(line["result_checksum"])
    assert len(sums) == 1


@pytest.mark.gpu
def test_example_encrypted_gpt2_lm_head_full():
    """gpt_model.cpp:883 logits at full size: 768 -> 50257 = 7 output ciphertexts sharing the 31 baby-step rotations (2.7 GB of
    diagonals over Q P, about a minute of host-side encoding): every one of the 50257 decrypted logits equals W x mod t."""
    out = subprocess.run([build_example("encrypted_gpt2_linear"), "lm_head", "1"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "OK" in out.stdout and "7 output ciphertext" in out.stdout and "MISMATCH" not in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [0, 1, 2])
def test_example_transformer_block_skeleton(ranks):
    """BASELINE configs[4] as one object: QKV (gpt_model.cpp:793) -> v hand-over (attention over one position) -> attention output
    projection + residual -> FFN up / down (:848) + residual, chained on the device for 4 tokens at N=8192, 5 data limbs + special
    prime.  Every stage decrypts to the plaintext result and the noise budget stays positive after each layer.  ranks = 1: the
    multi-process token-sharded driver (fork, RCCL id through a file, all-gather of the output ciphertexts) at world size 1; ranks = 2:
    the same slices played by one process on this box's single GPU."""
    import json
    out = subprocess.run([build_example("encrypted_gpt2_block"), "4", "1", "json", str(ranks)], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(next(l for l in out.stdout.splitlines() if l.startswith("{")))
    assert line["correct"] and line["stage_mismatches"] == [0, 0, 0, 0, 0] and line["tokens"] == 4 and line["ranks"] == ranks
    b = line["noise_budget_bits"]
    assert b["fresh"] > b["qkv"] > b["v_handover"] > b["h1"] > b["ffn_up_handover"] > b["h2"] > 0, b
    assert line["key_switches_per_token"] >= 62 * 4


@pytest.mark.gpu
def test_example_two_transformer_blocks_fit_the_five_limb_budget():
    """Two chained blocks (ten multiplicative levels, 510 key switches per token) still decrypt to the plaintext result at N=8192 with
    five 60-bit data limbs: the remaining budget is small but positive - the honest depth of this parameter set."""
    import json
    out = subprocess.run([build_example("encrypted_gpt2_block"), "2", "1", "json", "0", "2"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(next(l for l in out.stdout.splitlines() if l.startswith("{")))
    assert line["correct"] and line["layers"] == 2 and 0 < line["noise_budget_bits"]["gathered_output"] < 60


@pytest.mark.gpu
def test_bench_multi_gpu_programs_at_world_size_one():
    """What bench.py's rank 0 runs after the timed region at N > 1 (the C++ multi-process programs for configs[3] and configs[4]), here with
    one rank per program: both must report their own correctness checks and parse into the bench line's entries."""
    import bench
    r = bench.multi_gpu_programs(1)
    assert r["configs3_cpp_host"].get("matches_world1_recomputation") is True, r
    assert r["configs3_cpp_host"]["world"] == 1 and r["configs3_cpp_host"]["ct_mul_per_s"] > 0
    blk = r["configs4_token_sharded_block"]
    assert blk.get("correct") is True and blk["ranks"] == 1 and blk["tokens"] == 8 and blk["stage_mismatches"] == [0, 0, 0, 0, 0], r
    assert r["configs4_single_token_tensor_parallel_ffn"].get("correct") is True and r["configs4_single_token_tensor_parallel_ffn"]["world"] == 1, r
