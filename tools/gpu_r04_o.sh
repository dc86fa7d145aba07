#!/bin/bash
# after the branch-free plaintext-product kernel: the layers at 8 tokens, the stack, the full GPU suite
mkdir -p gpurun_out/r04o
timeout 600 ./examples/encrypted_gpt2_linear all 3 json 8 > gpurun_out/r04o/layers_t8.txt 2>&1; echo "layers rc=$?"
timeout 600 ./examples/encrypted_gpt2_linear all 5 json 1 > gpurun_out/r04o/layers_t1.txt 2>&1; echo "layers t1 rc=$?"
python - <<'PY'
import json
for f in ("layers_t8", "layers_t1"):
    for l in open(f"gpurun_out/r04o/{f}.txt"):
        if l.startswith("{"):
            d = json.loads(l); print(f, d.get("layer"), d.get("ms_per_token"), d.get("correct"))
PY
timeout 600 ./examples/encrypted_gpt2_block_act 8 2 json ladder 2>&1 | tail -2 | cut -c1-500
timeout 600 ./examples/encrypted_gpt2_stack 8 2 json 10 2>&1 | tail -2 | cut -c1-700
timeout 1300 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee gpurun_out/r04o/pytest.txt
