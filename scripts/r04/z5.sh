#!/bin/bash
# chain tails: do two verifier wavefronts keep up with fourteen producers now?
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z5; mkdir -p $O
KEY8_VARIANTS=12,14,14n timeout 300 python scripts/key8_ab.py 2>&1 | tail -2 | tee $O/tails_roles_ab.jsonl
