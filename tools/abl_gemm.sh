# Timing-only ablations of beaver_gemm_ws_kernel (results wrong on purpose). Needs a library built with `make clean && make ABLATION=1`.
# DBG bits: 1 no E loads, 2 no PRNG, 4 no MFMA, 8 no limb split / LDS writes, 16 no B copy (27 = consumers alone).
for d in ${ABL:-0 1 2 4 27}; do echo "DBG=$d"; COGNN_GEMM_DBG=$d timeout -k 10 60 python tools/microbench.py gemm --iters 50 2>&1 | grep "raw" | head -1; done
