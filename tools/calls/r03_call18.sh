#!/bin/bash
# Round-3 GPU call 18: dlogits without the 1.4 GB zero fill (ops.lmhead_logp_bwd): the tests that reach the LM head backward.
set -u
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py tests/test_trainer_semantics_gpu.py tests/test_lora_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
