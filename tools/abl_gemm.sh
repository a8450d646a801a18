for d in ${ABL:-0 1 2 4 27}; do echo "DBG=$d"; COGNN_GEMM_DBG=$d timeout -k 10 60 python tools/microbench.py gemm --iters 50 2>&1 | grep "raw" | head -1; done
