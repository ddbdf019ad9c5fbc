#!/bin/bash
set -u
timeout 600 python tools/many_contexts.py 10 2>&1 | grep "context"
