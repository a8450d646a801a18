#!/bin/bash
# Runs ON THE GPU BOX with an ABLATION build (make ABLATION=1): timing-only variants of the one-column-tile grouped Beaver product
# (results wrong).  bits: 1 no operand loads, 2 no PRNG, 4 no MFMA, 8 no limb split of the mask, 16 no LDS reads of B, 32 no epilogue
for d in 0 1 2 4 8 16 10 11 27 20 32; do
  echo -n "DBG=$d  "; COGNN_GEMM_DBG=$d python3 tools/group_probe.py ${1:-131072} ${2:-128} ${3:-16} ${4:-8} 2>/dev/null | grep "us per call"
done
