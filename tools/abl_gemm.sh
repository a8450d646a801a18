# Timing-only ablations of the fused Beaver GEMM kernels (results wrong on purpose). Needs a library built with `make clean && make ABLATION=1`.
# DBG bits: 1 no E loads, 2 no PRNG, 4 no MFMA, 8 no limb split (/ LDS writes), 16 no B copy / fragment reads (27 = MFMA alone).
# COGNN_GEMM_NO_D16N=1 selects the wave-specialised kernel for the N = 64 shape, otherwise the register-direct one is measured.
for d in ${ABL:-0 1 2 4 8 10 11 16 27}; do echo "DBG=$d"; COGNN_GEMM_DBG=$d timeout -k 10 60 python tools/microbench.py gemm --iters 50 2>&1 | grep "one E stream" | head -1; done
