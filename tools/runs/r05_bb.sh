#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "many_clients or post_chain_bit_exact" 2>&1 | tail -2
