#!/bin/bash
# What bounds the self-attention forward kernel: timing builds with one ingredient removed each (csrc/nn_attention.hip GD_ATTN_ABLATE;
# results are wrong, times are what is asked), B16 H5 S4096 and B16 H10 S1024.   usage (GPU box): tools/attn_ablate.sh
cd $(dirname $0)/..
bash tools/nn_variants.sh "attn1:-DGD_ATTN_ABLATE=1" "attn2:-DGD_ATTN_ABLATE=2" "attn3:-DGD_ATTN_ABLATE=3" "attn4:-DGD_ATTN_ABLATE=4" "attn5:-DGD_ATTN_ABLATE=5" "attn6:-DGD_ATTN_ABLATE=6" > /dev/null 2>&1
echo "product:"; ATTN_SHAPES=2 python tools/attn_xcd_ab.py 2>/dev/null | tail -2
for v in "1 no exp2" "2 no LDS-DMA after the prologue" "3 no MFMA" "4 no LDS fragment reads" "5 no running-maximum pass" "6 no per-tile wait + barrier"; do
  set -- $v; echo "without: ${v#* }"; GD_NN_LIB=$PWD/ablate/libgd_nn_attn$1.so ATTN_SHAPES=2 python tools/attn_xcd_ab.py 2>/dev/null | tail -2
done
