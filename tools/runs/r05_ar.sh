#!/bin/bash
set -u
O=gpurun_out/r05ar; mkdir -p $O
timeout 600 python tools/soak_post.py 64 256 40 3 2>&1 | tail -4
timeout 600 python tools/soak_post.py 16 512 24 3 2>&1 | tail -4
timeout 900 python tools/soak_post.py 256 128 30 2 2>&1 | tail -3
